// stem.hip -- the ResNet stem in ONE kernel: conv 7x7 / stride 2 / pad 3 (3 -> 64) + folded BN + ReLU + zero-pad +
// MaxPool 3x3 / stride 2.
//
// What it replaces: minddet/models/centernet/src/resnet.py:199-204 + :226-233 (conv1 -> bn1 -> relu -> pad -> maxpool),
// the same stem every ResNet backbone of the reference graphs starts with.
//
// Why a dedicated kernel (r01 measurements): as two generic launches the stem costs 0.9 ms + 0.33 ms per 32-image
// step -- the implicit-GEMM conv stages every input pixel 49 times (once per tap, 16 B per LDS-DMA lane) and the
// 1.1 GB conv output makes a round trip through HBM just to be pooled 4:1.
//
// Layout (chosen for this kernel): the image arrives as [N, H + 16, W + 16, 4] bf16 -- 3 normalised channels + one
// zero, a zero border of 7 pixels left / top (3 conv + 2 pooling-halo + 2 alignment) and 9 right / bottom -- so
//   * two horizontally adjacent pixels are one 16-B chunk = two kx taps of the 7x7 window: the MFMA B fragment of a
//     K step (K = 16 = 4 taps x 4 channels) is ONE ds_read_b128 straight from the raw input patch in LDS
//     (no im2col staging at all), K = (ky, kx padded 7 -> 8, c) = 224;
//   * no bounds checks anywhere: padding taps read real zeros.
// A workgroup (4 waves) owns 4 x 16 pooled pixels: it loads the 23 x 72-pixel input patch (13 KiB) once, computes the
// 9 x 33 conv pixels that the pooling windows touch (10 MFMA pixel fragments of 32; the weights live in REGISTERS,
// 14 K steps x one 32-cout A fragment per wave), writes bias + ReLU -> bf16 into an LDS tile and pools it from there.
// Workgroups are persistent (grid-stride over tiles): weights are fetched once, and the next patch is DMAed while
// the current tile is in its epilogue.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

namespace md {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

constexpr int ST_PAD_LO = 7, ST_PAD_HI = 9;               // image border (pixels) in the input layout
constexpr int ST_TPH = 4, ST_TPW = 16;                    // pooled pixels per workgroup
constexpr int ST_CH = 2 * ST_TPH + 1, ST_CW = 2 * ST_TPW + 1;  // conv pixels the pooling windows touch: 9 x 33
constexpr int ST_NPX = ST_CH * ST_CW;                     // 297
constexpr int ST_NF = (ST_NPX + 31) / 32;                 // 10 pixel fragments
constexpr int ST_PR = 2 * ST_CH + 5, ST_PW = 2 * ST_CW + 6;    // input patch: 23 rows x 72 pixels
constexpr int ST_ROWB = ST_PW * 8;                        // 576 B per patch row
constexpr int ST_PATCH_BYTES = ST_PR * ST_ROWB;           // 13,248
constexpr int ST_PATCH_CHUNKS = ST_PATCH_BYTES / 16;      // 828
constexpr int ST_PATCH_DMAS = (ST_PATCH_CHUNKS + 63) / 64;  // 13 wave instructions
constexpr int ST_PATCH_ALLOC = ST_PATCH_DMAS * 1024;      // 13,312
constexpr int ST_TROW = 128 + 16;                         // conv tile row stride: 64 cout bf16 + 16 B (bank spread)
constexpr int ST_TILE_BYTES = ST_NPX * ST_TROW;           // conv tile [pixel][64 cout] bf16: 42,768
constexpr int ST_K = 224, ST_KSTEPS = ST_K / 16;          // 14
constexpr int ST_LDS = ST_PATCH_ALLOC + ST_TILE_BYTES;    // 56,080 B: two workgroups per CU

struct StemArgs {
    const uint16_t *x;     // [N, Hp, Wp, 4]
    const uint16_t *w;     // [64][224]
    const float *bias;     // [64]
    uint16_t *y;           // [N, H/4, W/4, 64]
    int N, Hp, Wp, Hq, Wq;  // padded input dims, pooled output dims
    int tiles_x, tiles_y, n_tiles;
    unsigned x_bytes;
};

__device__ __forceinline__ uint16_t st_f2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

__device__ __forceinline__ unsigned st_pk_bf16(float lo, float hi) {  // RNE, one instruction (gfx950)
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ unsigned pk_max_u16(unsigned x, unsigned y) {
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
    return r;
}

__global__ __launch_bounds__(256, 2) void stem_pool_kernel(StemArgs a) {
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *patch = smem, *tile = smem + ST_PATCH_ALLOC;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave >> 1, wp = wave & 1;  // cout half / pixel-fragment half
    const int lr = lane & 31, lh = lane >> 5;

    // ---- weights: this wave's 32 couts x 224 K as 14 A fragments in registers
    bf16x8 fa[ST_KSTEPS];
#pragma unroll
    for (int s = 0; s < ST_KSTEPS; ++s)
        fa[s] = *reinterpret_cast<const bf16x8 *>(a.w + (size_t)(wc * 32 + lr) * ST_K + s * 16 + lh * 8);
    float4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const float4 *>(a.bias + wc * 32 + 8 * g + 4 * lh);

    // ---- per-lane fragment geometry (the same for every tile)
    int b_off[ST_NF / 2];  // LDS byte offset of (conv pixel, kx 0 / 2 by lane half) inside the patch
    int t_off[ST_NF / 2];  // byte offset of the pixel's row in the conv tile, -1 past the 297 pixels
    int q_rc[ST_NF / 2];   // (row << 8) | col of the conv pixel inside the 9 x 33 region
#pragma unroll
    for (int f = 0; f < ST_NF / 2; ++f) {
        const int q = (wp * (ST_NF / 2) + f) * 32 + lr;
        const int qc = q < ST_NPX ? q : ST_NPX - 1;
        const int cyl = qc / ST_CW, cxl = qc - cyl * ST_CW;
        b_off[f] = (2 * cyl * ST_PW + 2 * cxl + 2 * lh) * 8;
        t_off[f] = q < ST_NPX ? q * ST_TROW : -1;
        q_rc[f] = (cyl << 8) | cxl;
    }
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    // patch DMA: wave w issues pieces w, w+4, ...; lane -> 16-B chunk i = piece * 64 + lane = (row, chunk in row)
    auto dma_patch = [&](int t) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, n = t / (a.tiles_x * a.tiles_y);
        const int iy0 = 4 * ty * ST_TPH + 2, ix0 = 4 * tx * ST_TPW + 2;  // padded coordinates of the patch origin
        const int base = ((n * a.Hp + iy0) * a.Wp + ix0) * 8;
#pragma unroll
        for (int j = 0; j < (ST_PATCH_DMAS + 3) / 4; ++j) {
            const int piece = wave + 4 * j;
            if (piece < ST_PATCH_DMAS) {
                const int i = piece * 64 + lane;
                const int row = i / (ST_ROWB / 16), ch = i - row * (ST_ROWB / 16);
                const unsigned voff = i < ST_PATCH_CHUNKS ? (unsigned)(base + row * a.Wp * 8 + ch * 16) : 0x80000000u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(patch + piece * 1024), 16, (int)voff, 0, 0, 0);
            }
        }
    };

    int t = blockIdx.x;
    if (t < a.n_tiles) dma_patch(t);
    for (; t < a.n_tiles; t += gridDim.x) {
        const int tx = t % a.tiles_x, ty = (t / a.tiles_x) % a.tiles_y, n = t / (a.tiles_x * a.tiles_y);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // patch landed; the previous tile's pooling reads are done

        f32x16 acc[ST_NF / 2];
#pragma unroll
        for (int f = 0; f < ST_NF / 2; ++f)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[f][e] = 0.f;
#pragma unroll
        for (int s = 0; s < ST_KSTEPS; ++s) {
            const int koff = (s >> 1) * ST_ROWB + (s & 1) * 32;  // (ky, kx0 = 0 | 4) of this K step
#pragma unroll
            for (int f = 0; f < ST_NF / 2; ++f) {
                const bf16x8 fb = *reinterpret_cast<const bf16x8 *>(patch + b_off[f] + koff);
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s], fb, acc[f], 0, 0, 0);
            }
        }
        __syncthreads();  // every wave has finished reading the patch
        if (t + (int)gridDim.x < a.n_tiles) dma_patch(t + gridDim.x);  // flies during the epilogue + pooling

        // ---- bias + ReLU -> bf16 -> conv tile [pixel][cout]; conv pixels outside the image (row / column -1 of the
        // first tile row / column) are the zero padding of the pool
#pragma unroll
        for (int f = 0; f < ST_NF / 2; ++f) {
            if (t_off[f] < 0) continue;
            const bool pad = (ty == 0 && (q_rc[f] >> 8) == 0) || (tx == 0 && (q_rc[f] & 0xff) == 0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = fmaxf(acc[f][4 * g + 0] + bv[g].x, 0.f), v1 = fmaxf(acc[f][4 * g + 1] + bv[g].y, 0.f);
                float v2 = fmaxf(acc[f][4 * g + 2] + bv[g].z, 0.f), v3 = fmaxf(acc[f][4 * g + 3] + bv[g].w, 0.f);
                uint2 pk;
                pk.x = st_pk_bf16(v0, v1);
                pk.y = st_pk_bf16(v2, v3);
                if (pad) pk = make_uint2(0u, 0u);
                *reinterpret_cast<uint2 *>(tile + t_off[f] + (wc * 32 + 8 * g + 4 * lh) * 2) = pk;
            }
        }
        __syncthreads();

        // ---- 3x3 / stride-2 max over the tile: thread -> (pooled pixel, 16-channel chunk).  Post-ReLU bf16 values are
        // non-negative, so their bit patterns order like unsigned integers: packed 16-bit integer max.
        {
            const int pp = tid >> 2, chunk = tid & 3;
            const int ppy = pp / ST_TPW, ppx = pp - ppy * ST_TPW;
            u32x4 o0 = {0u, 0u, 0u, 0u}, o1 = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    // the thread's two 16-B pieces are channels [8 c, 8 c + 8) and [32 + 8 c, ..): the four threads of a pooled pixel then read four
                    // CONSECUTIVE 16-B slots per instruction (r04; with [16 c, 16 c + 16) they read every second slot and the pixels of a
                    // ds_read_b128 lane group, 288 B apart, collided 3-way: profiles/r03_lds_bank_conflict_survey.txt)
                    const char *src = tile + ((2 * ppy + dy) * ST_CW + 2 * ppx + dx) * ST_TROW + chunk * 16;
                    const u32x4 v0 = *reinterpret_cast<const u32x4 *>(src), v1 = *reinterpret_cast<const u32x4 *>(src + 64);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        o0[e] = pk_max_u16(o0[e], v0[e]);
                        o1[e] = pk_max_u16(o1[e], v1[e]);
                    }
                }
            const int py = ty * ST_TPH + ppy, px = tx * ST_TPW + ppx;
            uint16_t *dst = a.y + (((size_t)n * a.Hq + py) * a.Wq + px) * 64 + chunk * 8;
            __builtin_nontemporal_store(o0, reinterpret_cast<u32x4 *>(dst));
            __builtin_nontemporal_store(o1, reinterpret_cast<u32x4 *>(dst + 32));
        }
    }
}

}  // namespace md

using namespace md;

extern "C" int md_stem_layout_pad(int which) { return which == 0 ? ST_PAD_LO : ST_PAD_HI; }

extern "C" int md_stem_pool(MD_AOT_ARGS) {
    // in: x[N, H+16, W+16, 4] bf16 (zero border 7 / 9, channel 3 zero), w[64, 224] bf16 (K = ky, kx 0..7, c 0..3; kx 7 and c 3
    //     zero), bias[64] f32 ; out: y[N, H/4, W/4, 64] bf16.  H % 16 == 0, W % 64 == 0.
    if (nparam != 4) return MD_ERR_NPARAM;
    if (!params || !ndims || !shapes || !params[1] || !params[2]) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16") || !dtype_is(dtypes, 2, "float32") ||
        !dtype_is(dtypes, 3, "bfloat16"))
        return MD_ERR_ARG;
    if (ndims[0] != 4 || ndims[1] != 2 || ndims[3] != 4) return MD_ERR_ARG;
    if (shapes[0][3] != 4 || shapes[1][0] != 64 || shapes[1][1] != ST_K || numel(ndims, shapes, 2) != 64 || shapes[3][3] != 64)
        return MD_ERR_ARG;
    StemArgs a;
    a.N = (int)shapes[0][0]; a.Hp = (int)shapes[0][1]; a.Wp = (int)shapes[0][2];
    const int H = a.Hp - ST_PAD_LO - ST_PAD_HI, W = a.Wp - ST_PAD_LO - ST_PAD_HI;
    if (H <= 0 || W <= 0 || H % (4 * ST_TPH) || W % (4 * ST_TPW)) return MD_ERR_ARG;
    a.Hq = H / 4; a.Wq = W / 4;
    if (shapes[3][0] != a.N || shapes[3][1] != a.Hq || shapes[3][2] != a.Wq) return MD_ERR_ARG;
    if (a.N == 0) return MD_OK;
    if (!params[0] || !params[3]) return MD_ERR_ARG;
    const long long x_bytes = (long long)a.N * a.Hp * a.Wp * 8;
    if (x_bytes >= 0x7fff0000LL) return MD_ERR_SIZE;
    a.x = (const uint16_t *)params[0]; a.w = (const uint16_t *)params[1]; a.bias = (const float *)params[2];
    a.y = (uint16_t *)params[3];
    a.x_bytes = (unsigned)x_bytes;
    a.tiles_x = a.Wq / ST_TPW; a.tiles_y = a.Hq / ST_TPH;
    const long long n_tiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (n_tiles > 0x7fffffffLL) return MD_ERR_SIZE;
    a.n_tiles = (int)n_tiles;
    if (ensure_dyn_lds((const void *)stem_pool_kernel, ST_LDS) != MD_OK) return MD_ERR_HIP;
    const int grid = a.n_tiles < 256 * 2 ? a.n_tiles : 256 * 2;  // persistent: two workgroups per CU
    hipLaunchKernelGGL(stem_pool_kernel, dim3((unsigned)grid), dim3(256), ST_LDS, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}
