// stemconv.hip -- the FIRST convolution of the one-stage detectors (3 input channels, stride 2) as a dedicated kernel:
// YOLOv5 v6 stem conv 6x6 / 2 / pad 2 (3 -> 32 for yolov5s) and YOLOv8 stem conv 3x3 / 2 / pad 1 (3 -> 64 for yolov8l), + folded BN +
// SiLU (the model-build surface of configs/yolov5, configs/yolov8; the reference's stems are conv + BN + activation cells like
// minddet/models/centernet/src/resnet.py:199-204).
//
// Why (r02 profiles/r02_yolov5s_conv_layers.json): on the generic implicit-GEMM path the stem reads an 8-channel-padded image (2.7x the
// real bytes), stages every pixel once per tap through the per-lane K walk and runs at 2.2 TB/s of algorithmic bytes / 0.28 of its
// roofline: 0.19 ms of the 1.94 ms YOLOv5s step.  Same idea as stem.hip, without the pooling:
//   * input in the STEM LAYOUT [N, H + 16, W + 16, 4] bf16 (3 channels + one zero, zero border 7 pixels left / top and 9 right /
//     bottom: nn_ops.to_stem_layout, or md_image_preprocess writes it directly): two horizontally adjacent pixels are one 16-B chunk =
//     two kx taps, so the MFMA B fragment of a K step (16 = 4 kx taps x 4 channels) is ONE ds_read_b128 straight from the raw input
//     patch in LDS (no im2col), padding taps read real zeros (no bounds checks), K = (ky, kx padded to a multiple of 4, c);
//   * the 6-tap window starts at an odd column of the 16-B chunking, so it runs as an 8-tap window one column early with zero weights
//     at both ends (K = 6 x 8 x 4 = 192); the 3-tap window is padded to 4 (K = 48);
//   * a workgroup (4 waves) owns 8 x 32 output pixels: patch (2*8 + KH - 2) x 72 pixels by LDS-DMA, weights in REGISTERS (KH * KX / 4 A
//     fragments per wave), bias + activation -> bf16 tile in LDS -> whole 16-B NHWC stores; workgroups are persistent and the next
//     patch is DMAed during the epilogue.
// HBM traffic per image: 4-channel input once (+ halo) + the output -- 0.31 GB per 32-image YOLOv5s step instead of 0.42 GB.
// Measured r02 (tools/ab_env_bench.sh MD_STEM_LAYOUT, batch 32, same box): YOLOv5s 16 440 -> 17 400 images/s (+5.9 %), YOLOv8l 4 304 -> 4 354 (+1.2 %).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

namespace md {

typedef __attribute__((ext_vector_type(8))) short sc_bf16x8;
typedef __attribute__((ext_vector_type(16))) float sc_f32x16;
typedef __attribute__((ext_vector_type(4))) float sc_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int sc_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int sc_u32x2;

constexpr int SC_TH = 8, SC_TW = 32;            // output pixels per workgroup: fragment f = tile row f, lane = column
constexpr int SC_PW = 72, SC_ROWB = SC_PW * 8;  // patch row: 72 pixels x 4 channels x 2 B = 576 B

struct StemConvArgs {
    const uint16_t *x;     // [N, Hp, Wp, 4]
    const uint16_t *w;     // [COUT][KH * KX * 4]
    const float *bias;     // [COUT]
    uint16_t *y;           // [N, Ho, Wo, COUT]
    int N, Hp, Wp, Ho, Wo, act;   // act: 0 none, 1 ReLU, 2 SiLU
    int tiles_x, tiles_y, n_tiles;
    unsigned x_bytes;
};

__device__ __forceinline__ unsigned sc_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float sc_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }   // as conv.hip

// KH 6: the 6x6 / pad 2 window as 6 rows x 8 columns starting at padded (2 oy + 5, 2 ox + 4); KH 3: 3x3 / pad 1 as 3 rows x 4 columns
// starting at padded (2 oy + 6, 2 ox + 6)
template <int KH, int COUT>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(StemConvArgs a) {
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int KX = KH == 6 ? 8 : 4, KSTEPS = KH * KX / 4, KTOT = KSTEPS * 16;
    constexpr int ROW0 = KH == 6 ? 5 : 6, COL0 = KH == 6 ? 4 : 6;       // patch origin offset (padded coordinates) of output pixel (0, 0)
    constexpr int PR = 2 * SC_TH + KH - 2;                               // patch rows
    constexpr int PATCH_CHUNKS = PR * SC_ROWB / 16, PATCH_DMAS = (PATCH_CHUNKS + 63) / 64, PATCH_ALLOC = PATCH_DMAS * 1024;
    constexpr int WCN = COUT / 32, WPN = 4 / WCN, NF = SC_TH / WPN;       // waves along cout / along pixel fragments, fragments per wave
    constexpr int TROW = COUT * 2 + 16;                                  // tile row stride [pixel][COUT] bf16 + 16 B (bank spread)
    constexpr int CPP = COUT / 8;                                        // 16-B chunks per output pixel
    static_assert(COUT == 32 || COUT == 64, "one or two 32-cout fragments");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *patch = smem, *tile = smem + PATCH_ALLOC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave / WPN, wp = wave % WPN;
    const int lr = lane & 31, lh = lane >> 5;

    sc_bf16x8 fa[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) fa[s] = *reinterpret_cast<const sc_bf16x8 *>(a.w + (size_t)(wc * 32 + lr) * KTOT + s * 16 + lh * 8);
    sc_f32x4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const sc_f32x4 *>(a.bias + wc * 32 + 8 * g + 4 * lh);

    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    auto dma_patch = [&](int t) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, n = t / (a.tiles_x * a.tiles_y);
        const int base = ((n * a.Hp + 2 * ty * SC_TH + ROW0) * a.Wp + 2 * tx * SC_TW + COL0) * 8;
#pragma unroll
        for (int j = 0; j < (PATCH_DMAS + 3) / 4; ++j) {
            const int piece = wave + 4 * j;
            if (piece < PATCH_DMAS) {
                const int i = piece * 64 + lane;
                const int row = i / (SC_ROWB / 16), ch = i - row * (SC_ROWB / 16);
                // the last patch columns of the right-most tile / rows of the bottom tile can lie past the tensor: the range check returns 0
                const unsigned voff = i < PATCH_CHUNKS ? (unsigned)(base + row * a.Wp * 8 + ch * 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(patch + piece * 1024), 16, (int)voff, 0, 0, 0);
            }
        }
    };

    int t = blockIdx.x;
    if (t < a.n_tiles) dma_patch(t);
    for (; t < a.n_tiles; t += gridDim.x) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, n = t / (a.tiles_x * a.tiles_y);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // patch landed; the previous tile's read-out is done

        sc_f32x16 acc[NF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int ky = s / (KX / 4), kx0 = 4 * (s % (KX / 4));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int cy = wp * NF + f;   // tile row of this fragment
                const sc_bf16x8 fb = *reinterpret_cast<const sc_bf16x8 *>(patch + ((2 * cy + ky) * SC_PW + 2 * lr + kx0 + 2 * lh) * 8);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], fb, acc[f], 0, 0, 0);
            }
        }
        __syncthreads();   // every wave has finished reading the patch
        if (t + (int)gridDim.x < a.n_tiles) dma_patch(t + gridDim.x);   // flies during the epilogue

#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int q = (wp * NF + f) * 32 + lr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = acc[f][4 * g + 0] + bv[g].x, v1 = acc[f][4 * g + 1] + bv[g].y;
                float v2 = acc[f][4 * g + 2] + bv[g].z, v3 = acc[f][4 * g + 3] + bv[g].w;
                if (a.act == 2) { v0 = sc_silu(v0); v1 = sc_silu(v1); v2 = sc_silu(v2); v3 = sc_silu(v3); }
                else if (a.act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                sc_u32x2 pk;
                pk.x = sc_pk_bf16(v0, v1);
                pk.y = sc_pk_bf16(v2, v3);
                *reinterpret_cast<sc_u32x2 *>(tile + q * TROW + (wc * 32 + 8 * g + 4 * lh) * 2) = pk;
            }
        }
        __syncthreads();
        // whole 16-B NHWC stores: CPP lanes per pixel
#pragma unroll
        for (int it = 0; it < SC_TH * SC_TW * CPP / 256; ++it) {
            const int e = tid + 256 * it, q = e / CPP, cc = e % CPP;
            const sc_u32x4 v = *reinterpret_cast<const sc_u32x4 *>(tile + q * TROW + cc * 16);
            const int oy = ty * SC_TH + (q >> 5), ox = tx * SC_TW + (q & 31);
            __builtin_nontemporal_store(v, reinterpret_cast<sc_u32x4 *>(a.y + (((size_t)n * a.Ho + oy) * a.Wo + ox) * COUT + cc * 8));
        }
    }
}

template <int KH, int COUT>
static int launch_stem_conv(StemConvArgs &a, hipStream_t s) {
    constexpr int KX = KH == 6 ? 8 : 4;
    (void)KX;
    constexpr int PR = 2 * SC_TH + KH - 2;
    constexpr int PATCH_ALLOC = ((PR * SC_ROWB / 16 + 63) / 64) * 1024;
    const int lds = PATCH_ALLOC + SC_TH * SC_TW * (COUT * 2 + 16);
    auto k = stem_conv_kernel<KH, COUT>;
    if (lds > 64 * 1024 && ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    const int grid = a.n_tiles < 256 * 3 ? a.n_tiles : 256 * 3;   // persistent: up to three workgroups per CU
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

}  // namespace md

using namespace md;

// in : x[N, H+16, W+16, 4] bf16 (the stem layout: zero border 7 / 9, channel 3 zero), w[COUT, KH*KX*4] bf16 with K = (ky, kx', c):
//      kh = 6: kx' = kx + 1 in 0..7 (columns 0 and 7 zero), kh = 3: kx' = kx in 0..3 (column 3 zero); c = 3 zero; bias[COUT] f32
// out: y[N, H/2, W/2, COUT] bf16, COUT = 32 or 64.   H % 16 == 0, W % 64 == 0.   extra: md_stem_conv_attrs {kh (6 | 3), act}
extern "C" int md_stem_conv(MD_AOT_ARGS) {
    if (nparam != 4) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || !params[1] || !params[2]) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16") || !dtype_is(dtypes, 2, "float32") ||
        !dtype_is(dtypes, 3, "bfloat16"))
        return MD_ERR_ARG;
    if (ndims[0] != 4 || ndims[1] != 2 || ndims[3] != 4) return MD_ERR_ARG;
    const md_stem_conv_attrs *at = (const md_stem_conv_attrs *)extra;
    if ((at->kh != 6 && at->kh != 3) || at->act < 0 || at->act > 2) return MD_ERR_ARG;
    const int64_t cout = shapes[3][3], ktot = at->kh == 6 ? 192 : 48;
    if ((cout != 32 && cout != 64) || shapes[0][3] != 4 || shapes[1][0] != cout || shapes[1][1] != ktot || numel(ndims, shapes, 2) != cout)
        return MD_ERR_ARG;
    StemConvArgs a;
    a.N = (int)shapes[0][0]; a.Hp = (int)shapes[0][1]; a.Wp = (int)shapes[0][2];
    const int H = a.Hp - 16, W = a.Wp - 16;
    if (H <= 0 || W <= 0 || H % (2 * SC_TH) || W % (2 * SC_TW)) return MD_ERR_ARG;
    a.Ho = H / 2; a.Wo = W / 2;
    if (shapes[3][0] != a.N || shapes[3][1] != a.Ho || shapes[3][2] != a.Wo) return MD_ERR_ARG;
    if (a.N == 0) return MD_OK;
    if (!params[0] || !params[3]) return MD_ERR_ARG;
    const long long x_bytes = (long long)a.N * a.Hp * a.Wp * 8;
    if (x_bytes >= 0x7fff0000LL) return MD_ERR_SIZE;
    a.x = (const uint16_t *)params[0]; a.w = (const uint16_t *)params[1]; a.bias = (const float *)params[2];
    a.y = (uint16_t *)params[3];
    a.x_bytes = (unsigned)x_bytes;
    a.act = at->act;
    a.tiles_x = a.Wo / SC_TW; a.tiles_y = a.Ho / SC_TH;
    const long long n_tiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (n_tiles > 0x7fffffffLL) return MD_ERR_SIZE;
    a.n_tiles = (int)n_tiles;
    hipStream_t s = (hipStream_t)stream;
    if (at->kh == 6) return cout == 32 ? launch_stem_conv<6, 32>(a, s) : launch_stem_conv<6, 64>(a, s);
    return cout == 32 ? launch_stem_conv<3, 32>(a, s) : launch_stem_conv<3, 64>(a, s);
}
