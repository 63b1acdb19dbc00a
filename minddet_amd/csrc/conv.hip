// conv.hip -- conv + folded-BN bias + (residual) + ReLU as an implicit GEMM on CDNA4 matrix cores.
//
// What it replaces: every nn.Conv2d -> nn.BatchNorm2d -> nn.ReLU chain (and the residual add)
// of the reference graphs, e.g. minddet/models/centernet/src/resnet.py:109-178 (BasicBlock /
// Bottleneck), :181-252 (ResNet), minddet/models/centerpoint/det3d_ms/models/necks/rpn.py:9-154,
// minddet/models/pointpillars/src/pointpillars.py:367-621; FC layers run as 1x1 convs.
//
// Layout (chosen for MI355X, not inherited): activations NHWC bf16 (Cin % 8 == 0, so one
// 16-byte load = 8 channels of one pixel, and 8 lanes = one 128-B line), weights
// [Cout_pad][Kpad] bf16 with K ordered (kh, kw, ci) and BN folded in, bias fp32.
//
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k].  The WEIGHTS are the MFMA A
// operand (rows) and the PIXELS the B operand (columns), so each lane ends up holding 4
// consecutive output channels of one pixel per accumulator group: the epilogue packs them to
// bf16x4 (ds_write_b64) into a [pixel][cout] LDS image and the tile leaves the CU as whole
// 16-byte / 128-B-line NHWC stores, with bias, residual and ReLU fused.
//
// Workgroup = 256 threads = 4 wave64; tile = (WC*FC*32 couts) x (WP*FP*32 pixels) x BK 64;
// v_mfma_f32_32x32x16_bf16; LDS tiles are [row][64 k] bf16 (128-B rows) with the 16-B chunk
// index XOR-swizzled by (row>>1)&7 so that both the ds_write_b128 staging stores and the
// ds_read_b128 fragment reads are bank-conflict free; double-buffered, one barrier per K tile,
// next tile's global loads issued before the MFMAs of the current one.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

namespace md {

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ConvArgs {
    const uint16_t *x;      // [N,H,W,Cin]
    const uint16_t *w;      // [Cout_pad][Kpad]
    const float *bias;      // [Cout_pad]
    const uint16_t *res;    // [N,Ho,Wo,Cout] or null
    uint16_t *y;            // [N,Ho,Wo,Cout]
    int N, H, W, Cin, Cout, Ho, Wo;
    int kh, kw, stride, pad, relu;
    int Kpad;     // padded K (multiple of 64)
    int Kreal;    // kh*kw*Cin
    int M;        // N*Ho*Wo
    int cpt;      // chunks (8 ch) per tap = Cin/8
    int n_ptiles, n_ctiles, pt_per_xcd;
};

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

constexpr int BK = 64;
constexpr int ROWB = BK * 2;  // bytes per LDS tile row

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int WC, int WP, int FC, int FP>
__global__ __launch_bounds__(256, 3) void conv_igemm_kernel(ConvArgs a) {
    constexpr int CT = WC * FC * 32, PT = WP * FP * 32;
    constexpr int A_ROWS = CT / 32, B_ROWS = PT / 32;  // 16-B chunks per thread per tile
    constexpr int TILE_BYTES = (CT + PT) * ROWB;
    constexpr int EP_STRIDE = CT * 2 + 16;  // epilogue image row stride (bytes)
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;
    // XCD-aware tile map (workgroups are dealt round-robin over the 8 XCDs, each with a private L2):
    // every XCD owns a CONTIGUOUS range of pixel tiles, and inside it the cout tile varies fastest, so the
    // workgroups that share an activation tile (and the 3x3 halo rows of its neighbours) hit the same L2.
    // Placement only affects speed: any dispatch order computes the same tiles.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = slot % a.n_ctiles, pt = xcd * a.pt_per_xcd + slot / a.n_ctiles;
    if (pt >= a.n_ptiles) return;
    const int cout0 = ct * CT, pix0 = pt * PT;

    const int chunk = tid & 7, row0 = tid >> 3;  // this thread stages rows row0 + 32*i, 16-B chunk `chunk`

    // per-thread pixel rows of the B (activation) tile
    int p_hw0[B_ROWS];   // packed (hi0 << 16) | (wi0 & 0xffff), hi0/wi0 = top-left input coord of the window
    int p_base[B_ROWS];  // n*H*W (pixel index), or -1 if the row is past M
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
        const int m = pix0 + row0 + 32 * i;
        if (m < a.M) {
            const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
            const int ho = r / a.Wo, wo = r - ho * a.Wo;
            const int hi0 = ho * a.stride - a.pad, wi0 = wo * a.stride - a.pad;
            p_hw0[i] = (hi0 << 16) | (wi0 & 0xffff);
            p_base[i] = n * a.H * a.W;
        } else {
            p_hw0[i] = 0;
            p_base[i] = -1;
        }
    }
    // K position of this thread's chunk: tap index and channel-chunk within tap, advanced by 8 chunks per tile
    int q_tap = chunk / a.cpt, q_cc = chunk - q_tap * a.cpt;
    int q_kh = q_tap / a.kw, q_kw = q_tap - q_kh * a.kw;
    const int n_taps = a.kh * a.kw;

    u32x4 ra[A_ROWS], rb[B_ROWS];
    const int nk = a.Kpad / BK;

    auto load_tile = [&](int kt) {
        // weights: plain 2-D, always in bounds (padded at pack time)
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int r = cout0 + row0 + 32 * i;
            ra[i] = *reinterpret_cast<const u32x4 *>(a.w + (size_t)r * a.Kpad + kt * BK + chunk * 8);
        }
        const bool tap_ok = q_tap < n_taps;
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            const int hi = (p_hw0[i] >> 16) + q_kh, wi = (int)(short)(p_hw0[i] & 0xffff) + q_kw;
            const bool ok = tap_ok && p_base[i] >= 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) {
                const size_t off = ((size_t)(p_base[i] + hi * a.W + wi)) * a.Cin + q_cc * 8;
                v = *reinterpret_cast<const u32x4 *>(a.x + off);
            }
            rb[i] = v;
        }
        // advance this thread's K position by one tile (8 chunks)
        q_cc += 8;
        while (q_cc >= a.cpt) {
            q_cc -= a.cpt;
            ++q_tap;
            if (++q_kw == a.kw) { q_kw = 0; ++q_kh; }
        }
    };
    auto store_tile = [&](int buf) {
        char *A = smem + buf * TILE_BYTES, *B = A + CT * ROWB;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<u32x4 *>(A + swz(row0 + 32 * i, chunk)) = ra[i];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<u32x4 *>(B + swz(row0 + 32 * i, chunk)) = rb[i];
    };

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int lr = lane & 31, lh = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const char *A = smem + buf * TILE_BYTES, *B = A + CT * ROWB;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 fa[FC], fb[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) {
                const int r = (wc * FC + i) * 32 + lr;
                fa[i] = *reinterpret_cast<const bf16x8 *>(A + swz(r, kk * 2 + lh));
            }
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int r = (wp * FP + j) * 32 + lr;
                fb[j] = *reinterpret_cast<const bf16x8 *>(B + swz(r, kk * 2 + lh));
            }
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  Residual rows are fetched first so that their HBM latency overlaps the LDS transpose.
    constexpr int CPP = CT / 8;                 // 16-B chunks per pixel row of the tile
    constexpr int EP_ITERS = PT * CPP / 256;    // chunks per thread
    u32x4 rres[EP_ITERS];
    if (a.res) {
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int e = tid + it * 256;
            const int p_local = e / CPP, cc = e % CPP;
            const int m = pix0 + p_local, c = cout0 + cc * 8;
            rres[it] = (u32x4){0u, 0u, 0u, 0u};
            if (m < a.M && c < a.Cout) rres[it] = *reinterpret_cast<const u32x4 *>(a.res + (size_t)m * a.Cout + c);
        }
    }
    // bias (+ReLU when no residual) -> bf16x4 -> LDS [pixel][cout] image
    char *E = smem;
#pragma unroll
    for (int i = 0; i < FC; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c_local = (wc * FC + i) * 32 + 8 * g + 4 * lh;  // 4 consecutive couts
            const float4 bv = *reinterpret_cast<const float4 *>(a.bias + cout0 + c_local);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int p_local = (wp * FP + j) * 32 + lr;
                float v0 = acc[i][j][4 * g + 0] + bv.x, v1 = acc[i][j][4 * g + 1] + bv.y;
                float v2 = acc[i][j][4 * g + 2] + bv.z, v3 = acc[i][j][4 * g + 3] + bv.w;
                if (a.relu && !a.res) {
                    v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                }
                uint2 pk;
                pk.x = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);
                pk.y = (unsigned)f2bf(v2) | ((unsigned)f2bf(v3) << 16);
                *reinterpret_cast<uint2 *>(E + p_local * EP_STRIDE + c_local * 2) = pk;
            }
        }
    }
    __syncthreads();
    // ---- coalesced NHWC store: 16 B (8 couts) per lane, CT/8 lanes per pixel
#pragma unroll
    for (int it = 0; it < EP_ITERS; ++it) {
        const int e = tid + it * 256;
        const int p_local = e / CPP, cc = e % CPP;
        const int m = pix0 + p_local, c = cout0 + cc * 8;
        if (m >= a.M || c >= a.Cout) continue;
        u32x4 v = *reinterpret_cast<const u32x4 *>(E + p_local * EP_STRIDE + cc * 16);
        const size_t off = (size_t)m * a.Cout + c;
        if (a.res) {
            const u32x4 rv = rres[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float lo = bf2f((uint16_t)(v[k] & 0xffff)) + bf2f((uint16_t)(rv[k] & 0xffff));
                float hi = bf2f((uint16_t)(v[k] >> 16)) + bf2f((uint16_t)(rv[k] >> 16));
                if (a.relu) { lo = fmaxf(lo, 0.f); hi = fmaxf(hi, 0.f); }
                v[k] = (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16);
            }
        }
        *reinterpret_cast<u32x4 *>(a.y + off) = v;
    }
}

template <int WC, int WP, int FC, int FP>
static int launch_conv(ConvArgs &a, hipStream_t s) {
    constexpr int CT = WC * FC * 32, PT = WP * FP * 32;
    a.n_ctiles = (a.Cout + CT - 1) / CT;
    a.n_ptiles = (a.M + PT - 1) / PT;
    // one staging buffer is enough when the whole K fits one tile (1x1 convs on 64 channels): more
    // workgroups per CU for the HBM-bound layers
    const int nbuf = a.Kpad / BK > 1 ? 2 : 1;
    const int tile_bytes = (CT + PT) * ROWB * nbuf;
    constexpr int ep_bytes = PT * (CT * 2 + 16);
    const int lds = tile_bytes > ep_bytes ? tile_bytes : ep_bytes;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    const long long blocks = (long long)a.n_ctiles * a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    auto k = conv_igemm_kernel<WC, WP, FC, FP>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return MD_ERR_HIP;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

}  // namespace md

using namespace md;

// Required weight padding for a given Cout (the tile the dispatcher will pick): exported so the
// host packer pads consistently.
extern "C" int md_conv2d_cout_tile(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

extern "C" int md_conv2d(MD_AOT_ARGS) {
    if (nparam != 5) return MD_ERR_NPARAM;
    if (!params || !extra || !params[0] || !params[1] || !params[2] || !params[4]) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16") || !dtype_is(dtypes, 2, "float32") ||
        !dtype_is(dtypes, 4, "bfloat16"))
        return MD_ERR_ARG;
    if (!ndims || !shapes || ndims[0] != 4 || ndims[1] != 2 || ndims[4] != 4) return MD_ERR_ARG;
    const md_conv2d_attrs *at = (const md_conv2d_attrs *)extra;
    ConvArgs a;
    a.x = (const uint16_t *)params[0];
    a.w = (const uint16_t *)params[1];
    a.bias = (const float *)params[2];
    a.res = (const uint16_t *)params[3];
    a.y = (uint16_t *)params[4];
    a.N = (int)shapes[0][0]; a.H = (int)shapes[0][1]; a.W = (int)shapes[0][2]; a.Cin = (int)shapes[0][3];
    a.Ho = (int)shapes[4][1]; a.Wo = (int)shapes[4][2]; a.Cout = (int)shapes[4][3];
    a.kh = at->kh; a.kw = at->kw; a.stride = at->stride; a.pad = at->pad; a.relu = at->relu;
    if (a.kh < 1 || a.kw < 1 || a.stride < 1 || a.pad < 0) return MD_ERR_ARG;
    if (a.Cin % 8 || a.Cout % 8 || shapes[4][0] != a.N) return MD_ERR_ARG;
    if (a.Ho != (a.H + 2 * a.pad - a.kh) / a.stride + 1 || a.Wo != (a.W + 2 * a.pad - a.kw) / a.stride + 1)
        return MD_ERR_ARG;
    a.Kreal = a.kh * a.kw * a.Cin;
    a.Kpad = (int)shapes[1][1];
    const int ctile = md_conv2d_cout_tile(a.Cout);
    const int cout_pad = (a.Cout + ctile - 1) / ctile * ctile;
    if (a.Kpad % BK || a.Kpad < a.Kreal || shapes[1][0] != cout_pad) return MD_ERR_ARG;
    if (numel(ndims, shapes, 2) != cout_pad) return MD_ERR_ARG;
    if (params[3] && (ndims[3] != 4 || numel(ndims, shapes, 3) != numel(ndims, shapes, 4))) return MD_ERR_ARG;
    const long long M = (long long)a.N * a.Ho * a.Wo;
    if (M <= 0) return MD_OK;
    if (M > 0x7fffffffLL || (long long)a.N * a.H * a.W > 0x7fffffffLL / 2 || a.H > 32000 || a.W > 32000)
        return MD_ERR_SIZE;
    a.M = (int)M;
    a.cpt = a.Cin / 8;
    hipStream_t s = (hipStream_t)stream;
    if (ctile == 128) return launch_conv<2, 2, 2, 2>(a, s);
    if (ctile == 64) return launch_conv<1, 4, 2, 2>(a, s);
    return launch_conv<1, 4, 1, 2>(a, s);
}
