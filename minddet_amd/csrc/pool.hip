// pool.hip -- HBM-bound NHWC bf16 helpers between the conv kernels: max-pool, FPN top-down
// (nearest upsample + lateral add), channel slice + cast.  All are pure streaming kernels:
// 16 bytes (8 channels) per lane, one 128-B line per 8 lanes, grid-stride.
//
// Reference counterparts: zero-pad + MaxPool2d(3,2) of the ResNet stem
// (minddet/models/centernet/src/resnet.py:199-204,247); the FPN top-down path and the channel
// slice have no reference counterpart -- "parity unpinned", checked against
// torch.nn.functional on the same bf16 values (max / add of two bf16 are exact or one rounding).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

namespace md {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ float pbf2f(unsigned v16) { return __uint_as_float(v16 << 16); }
__device__ __forceinline__ unsigned pf2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// max of two packed bf16 pairs (compare as floats; -0/+0 and NaN corner cases follow fmaxf)
__device__ __forceinline__ unsigned max_bf16x2(unsigned a, unsigned b) {
    const float alo = pbf2f(a & 0xffffu), blo = pbf2f(b & 0xffffu);
    const float ahi = pbf2f(a >> 16), bhi = pbf2f(b >> 16);
    const unsigned lo = alo >= blo ? (a & 0xffffu) : (b & 0xffffu);
    const unsigned hi = ahi >= bhi ? (a >> 16) : (b >> 16);
    return lo | (hi << 16);
}

// zero_pad != 0: the window's out-of-image taps contribute 0 (explicit zero Pad then MaxPool,
// resnet.py:199-204); zero_pad == 0: they are ignored (-inf padding, torch semantics).
__global__ void maxpool_nhwc_kernel(const uint16_t *__restrict__ x, uint16_t *__restrict__ y, int N, int H, int W,
                                    int C, int Ho, int Wo, int k, int stride, int pad, int zero_pad) {
    const int cv = C / 8;
    const size_t total = (size_t)N * Ho * Wo * cv;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(e % cv);
        size_t p = e / cv;
        const int wo = (int)(p % Wo); p /= Wo;
        const int ho = (int)(p % Ho);
        const int n = (int)(p / Ho);
        u32x4 best;
        bool have = false;
        bool touched_pad = false;
        for (int dy = 0; dy < k; ++dy) {
            const int hi = ho * stride - pad + dy;
            for (int dx = 0; dx < k; ++dx) {
                const int wi = wo * stride - pad + dx;
                if ((unsigned)hi >= (unsigned)H || (unsigned)wi >= (unsigned)W) { touched_pad = true; continue; }
                const u32x4 v = *reinterpret_cast<const u32x4 *>(x + (((size_t)n * H + hi) * W + wi) * C + c8 * 8);
                if (!have) { best = v; have = true; }
                else {
#pragma unroll
                    for (int q = 0; q < 4; ++q) best[q] = max_bf16x2(best[q], v[q]);
                }
            }
        }
        if (!have) best = (u32x4){0u, 0u, 0u, 0u};
        else if (zero_pad && touched_pad) {
#pragma unroll
            for (int q = 0; q < 4; ++q) best[q] = max_bf16x2(best[q], 0u);
        }
        *reinterpret_cast<u32x4 *>(y + e * 8) = best;
    }
}

// y[n,h,w,:] = lateral[n,h,w,:] + top[n, floor(h*Ht/H), floor(w*Wt/W), :]   (nearest, F.interpolate(size=))
__global__ void upsample_add_kernel(const uint16_t *__restrict__ lat, const uint16_t *__restrict__ top,
                                    uint16_t *__restrict__ y, int N, int H, int W, int C, int Ht, int Wt) {
    const int cv = C / 8;
    const size_t total = (size_t)N * H * W * cv;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(e % cv);
        size_t p = e / cv;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const int ht = min((int)(((long long)h * Ht) / H), Ht - 1), wt = min((int)(((long long)w * Wt) / W), Wt - 1);
        const u32x4 a = *reinterpret_cast<const u32x4 *>(lat + e * 8);
        const u32x4 b = *reinterpret_cast<const u32x4 *>(top + (((size_t)n * Ht + ht) * Wt + wt) * C + c8 * 8);
        u32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float lo = pbf2f(a[q] & 0xffffu) + pbf2f(b[q] & 0xffffu);
            const float hi = pbf2f(a[q] >> 16) + pbf2f(b[q] >> 16);
            asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(o[q]) : "v"(lo), "v"(hi));  // RNE pack, one instruction
        }
        *reinterpret_cast<u32x4 *>(y + e * 8) = o;
    }
}

// out[m, j] = float(in[m, c0 + j]) for j < cw
__global__ void slice_cast_kernel(const uint16_t *__restrict__ in, float *__restrict__ out, size_t M, int C, int c0,
                                  int cw) {
    const size_t total = M * cw;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t m = e / cw;
        const int j = (int)(e % cw);
        out[e] = pbf2f(in[m * C + c0 + j]);
    }
}

// out[n, j, h, w] = float(in[n, h, w, c0 + j]): the head tensors the reference decodes are NCHW fp32
// (centernet/src/decode.py:151-196); LDS-tiled transpose so both sides move whole lines.
__global__ __launch_bounds__(256) void nhwc_to_nchw_f32_kernel(const uint16_t *__restrict__ in, float *__restrict__ out,
                                                               int HW, int C, int c0, int cw) {
    __shared__ float tile[64][65];
    const int n = blockIdx.z, p0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int r = ty; r < 64; r += 4) {   // r: pixel within tile, tx: channel
        const int p = p0 + r, j = j0 + tx;
        tile[r][tx] = (p < HW && j < cw) ? pbf2f(in[((size_t)n * HW + p) * C + c0 + j]) : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 64; r += 4) {   // r: channel within tile, tx: pixel
        const int j = j0 + r, p = p0 + tx;
        if (j < cw && p < HW) out[((size_t)n * cw + j) * HW + p] = tile[tx][r];
    }
}

// dst[n,h,w,c0+c] = src[n,h/up,w/up,c]  (up = 1: channel-slice copy; up = 2: nearest upsample into a slice)
// (the source may itself be a channel slice [sc0, sc0 + C) of a tensor with Cs channels)
__global__ void slice_write_kernel(const uint16_t *__restrict__ src, uint16_t *__restrict__ dst, int N, int H, int W, int C,
                                   int Ctot, int c0, int up, int Cs, int sc0) {
    const int cv = C / 8;
    const size_t total = (size_t)N * H * W * cv;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(e % cv);
        size_t p = e / cv;
        const int w = (int)(p % W); p /= W;
        const int h = (int)(p % H);
        const int n = (int)(p / H);
        const int Hs = H / up, Ws = W / up;
        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + (((size_t)n * Hs + h / up) * Ws + w / up) * Cs + sc0 + c8 * 8);
        *reinterpret_cast<u32x4 *>(dst + (((size_t)n * H + h) * W + w) * Ctot + c0 + c8 * 8) = v;
    }
}

// ---- the pooling chain of an SPPF block (build-authored YOLOv5 / YOLOv8 of BASELINE configs[1], [3]; the reference names the families only) in
// ONE launch, in place on the block's concat buffer: with x = channels [0, C) of buf (the block's first conv writes them there),
//   buf[.., C:2C] = mp(x),  buf[.., 2C:3C] = mp(mp(x)),  buf[.., 3C:4C] = mp(mp(mp(x))),   mp = max-pool k x k / stride 1 / pad k/2 (-inf padding).
// With out-of-image taps ignored, mp o mp is the max over the (2k-1)^2 window and mp o mp o mp over (3k-2)^2 -- every in-image tap of the
// big window is reachable through an in-image centre -- and a square window's max is separable: rows first, then columns.  A workgroup
// holds one image x CC 16-B channel groups in LDS: x, then the three row maxima, then the column maxima of those go out.  bf16 values are
// mapped to order-preserving 16-bit keys (negative: all bits flipped, else the sign bit set) so that a max is one v_pk_max_u16 per pair.
// Replaces three md_maxpool2d launches (25 global loads per output each) and four concat copies: 103 -> 22 us on YOLOv5s' 32 x 20 x 20 x 256.
typedef unsigned short sp_u16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned sp_key(unsigned v) { return v ^ ((((v >> 15) & 0x00010001u) * 0x7fffu) | 0x80008000u); }
__device__ __forceinline__ unsigned sp_unkey(unsigned k) { return k ^ ((((~k >> 15) & 0x00010001u) * 0x7fffu) | 0x80008000u); }
// (whole-vector form: an element-by-element loop over u32x4 with a 2 x u16 bit cast per element was compiled into ONE v_pk_max_u16 whose
// result was broadcast to all four dwords -- hipcc 7.2, found by the bit-compare test)
__device__ __forceinline__ u32x4 sp_max(u32x4 a, u32x4 b) {
    return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(sp_u16x8, a), __builtin_bit_cast(sp_u16x8, b)));
}

__global__ __launch_bounds__(256) void sppf_pool_kernel(uint16_t *__restrict__ buf, int H, int W, int C, int Ctot, int R, int CC) {
    extern __shared__ __attribute__((aligned(16))) char sp_smem[];
    const int HW = H * W, items = HW * CC;
    u32x4 *a0 = reinterpret_cast<u32x4 *>(sp_smem), *a1 = a0 + items, *a2 = a1 + items, *a3 = a2 + items;
    const int groups = C / 8 / CC;
    const int n = blockIdx.x / groups, c0 = (blockIdx.x % groups) * CC;   // first 16-B channel group of this workgroup
    uint16_t *img = buf + (size_t)n * HW * Ctot + c0 * 8;
    for (int i = threadIdx.x; i < items; i += 256) {
        const int px = i / CC, ch = i - px * CC;
        u32x4 v = *reinterpret_cast<const u32x4 *>(img + (size_t)px * Ctot + ch * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = sp_key(v[q]);
        a0[i] = v;
    }
    __syncthreads();
    // (a tap beyond the image edge is replaced by the edge pixel, which is inside every window that reaches the edge: max is idempotent, so
    // that equals ignoring it -- and the loops have no branch, all their LDS reads are independent and issue back to back)
    for (int i = threadIdx.x; i < items; i += 256) {   // row maxima of radius R, 2R, 3R
        const int px = i / CC, c = px % W;
        u32x4 m = a0[i];
        for (int d = 1; d <= 3 * R; ++d) {
            m = sp_max(m, sp_max(a0[i - min(d, c) * CC], a0[i + min(d, W - 1 - c) * CC]));
            if (d == R) a1[i] = m;
            if (d == 2 * R) a2[i] = m;
        }
        a3[i] = m;
    }
    __syncthreads();
    const int rs = W * CC;   // one image row in items
    for (int i = threadIdx.x; i < items; i += 256) {   // column maxima of the row maxima -> the three outputs
        const int px = i / CC, ch = i - px * CC, r = px / W;
        u32x4 m1 = a1[i], m2 = a2[i], m3 = a3[i];
        for (int d = 1; d <= 3 * R; ++d) {
            const int up = min(d, r) * rs, dn = min(d, H - 1 - r) * rs;
            m3 = sp_max(m3, sp_max(a3[i - up], a3[i + dn]));
            if (d <= 2 * R) m2 = sp_max(m2, sp_max(a2[i - up], a2[i + dn]));
            if (d <= R) m1 = sp_max(m1, sp_max(a1[i - up], a1[i + dn]));
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { m1[q] = sp_unkey(m1[q]); m2[q] = sp_unkey(m2[q]); m3[q] = sp_unkey(m3[q]); }
        uint16_t *o = img + (size_t)px * Ctot + ch * 8;
        *reinterpret_cast<u32x4 *>(o + C) = m1;
        *reinterpret_cast<u32x4 *>(o + 2 * C) = m2;
        *reinterpret_cast<u32x4 *>(o + 3 * C) = m3;
    }
}

static inline unsigned grid_for(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 8192 ? 8192 : (b == 0 ? 1 : b));
}

}  // namespace md

using namespace md;

extern "C" int md_maxpool2d(MD_AOT_ARGS) {
    if (nparam != 2) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4 || ndims[1] != 4) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16")) return MD_ERR_ARG;
    const md_pool_attrs *at = (const md_pool_attrs *)extra;
    const int N = (int)shapes[0][0], H = (int)shapes[0][1], W = (int)shapes[0][2], C = (int)shapes[0][3];
    const int Ho = (int)shapes[1][1], Wo = (int)shapes[1][2];
    if (C % 8 || shapes[1][3] != C || shapes[1][0] != N || at->k < 1 || at->stride < 1 || at->pad < 0) return MD_ERR_ARG;
    if (Ho != (H + 2 * at->pad - at->k) / at->stride + 1 || Wo != (W + 2 * at->pad - at->k) / at->stride + 1)
        return MD_ERR_ARG;
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    if (total == 0) return MD_OK;
    hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (uint16_t *)params[1], N, H, W, C, Ho, Wo, at->k, at->stride, at->pad,
                       at->zero_pad);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

// channel groups (of 8 channels) per workgroup that md_sppf_pool would use for an H x W map with C channels, 0 = does not fit LDS
// (four [H*W][groups] arrays of 16 B; the caller then keeps the three md_maxpool2d launches).  Pure function, callable without a GPU.
extern "C" int md_sppf_pool_groups(int H, int W, int C) {
    if (H < 1 || W < 1 || C < 8 || C % 8) return 0;
    for (int cc = 4; cc >= 1; cc >>= 1)
        if ((C / 8) % cc == 0 && (long long)H * W * cc * 64 <= 160 * 1024) return cc;
    return 0;
}

// in/out: buf[N,H,W,Ctot] bf16, Ctot >= 4 C: reads channels [0, C), writes [C, 4C).   extra: md_sppf_attrs {C, k}
extern "C" int md_sppf_pool(MD_AOT_ARGS) {
    if (nparam != 1) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4 || !dtype_is(dtypes, 0, "bfloat16")) return MD_ERR_ARG;
    const md_sppf_attrs *at = (const md_sppf_attrs *)extra;
    const long long N = shapes[0][0], H = shapes[0][1], W = shapes[0][2], Ctot = shapes[0][3];
    const int C = at->channels, k = at->k;
    if (C < 8 || C % 8 || Ctot % 8 || Ctot < 4LL * C || k < 1 || k % 2 == 0) return MD_ERR_ARG;
    if (N * H * W == 0) return MD_OK;
    if (!params[0]) return MD_ERR_ARG;
    if (H > 4096 || W > 4096) return MD_ERR_SIZE;
    const int cc = md_sppf_pool_groups((int)H, (int)W, C);
    if (cc == 0 || N * (C / 8 / cc) > 0x7fffffffLL) return MD_ERR_SIZE;
    const int lds = (int)(H * W * cc * 64);
    if (ensure_dyn_lds((const void *)sppf_pool_kernel, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(sppf_pool_kernel, dim3((unsigned)(N * (C / 8 / cc))), dim3(256), lds, (hipStream_t)stream, (uint16_t *)params[0], (int)H, (int)W, C,
                       (int)Ctot, k / 2, cc);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_upsample_add(MD_AOT_ARGS) {
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !ndims || !shapes || ndims[0] != 4 || ndims[1] != 4 || ndims[2] != 4) return MD_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (!dtype_is(dtypes, i, "bfloat16")) return MD_ERR_ARG;
    const int N = (int)shapes[0][0], H = (int)shapes[0][1], W = (int)shapes[0][2], C = (int)shapes[0][3];
    const int Ht = (int)shapes[1][1], Wt = (int)shapes[1][2];
    if (C % 8 || shapes[1][0] != N || shapes[1][3] != C || Ht < 1 || Wt < 1) return MD_ERR_ARG;
    for (int d = 0; d < 4; ++d)
        if (shapes[2][d] != shapes[0][d]) return MD_ERR_ARG;
    const size_t total = (size_t)N * H * W * (C / 8);
    if (total == 0) return MD_OK;
    hipLaunchKernelGGL(upsample_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (const uint16_t *)params[1], (uint16_t *)params[2], N, H, W, C, Ht, Wt);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_nhwc_to_nchw_f32(MD_AOT_ARGS) {
    // in x[N,H,W,C] bf16 ; out y[N,width,H,W] f32 = x[..., c0:c0+width] transposed.  extra: md_slice_attrs
    if (nparam != 2) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4 || ndims[1] != 4) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32")) return MD_ERR_ARG;
    const md_slice_attrs *at = (const md_slice_attrs *)extra;
    const int N = (int)shapes[0][0], H = (int)shapes[0][1], W = (int)shapes[0][2], C = (int)shapes[0][3];
    if (at->c0 < 0 || at->width < 1 || at->c0 + at->width > C) return MD_ERR_ARG;
    if (shapes[1][0] != N || shapes[1][1] != at->width || shapes[1][2] != H || shapes[1][3] != W) return MD_ERR_ARG;
    if ((size_t)N * H * W == 0) return MD_OK;
    if (N > 65535) return MD_ERR_SIZE;
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((HW + 63) / 64, (at->width + 63) / 64, N), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t *)params[0], (float *)params[1], HW, C, at->c0, at->width);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

static int slice_write_impl(MD_AOT_ARGS, int up) {
    if (nparam != 2) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4 || ndims[1] != 4) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16")) return MD_ERR_ARG;
    // md_concat_copy: md_slice_attrs {c0, width}; md_upsample2x: md_upsample2x_attrs {c0, width, src_c0} (the source may be a channel slice)
    const md_upsample2x_attrs *at = (const md_upsample2x_attrs *)extra;
    const int sc0 = up == 2 ? at->src_c0 : 0;
    const int N = (int)shapes[0][0], Hs = (int)shapes[0][1], Ws = (int)shapes[0][2], Cs = (int)shapes[0][3];
    const int C = up == 2 ? at->width : Cs;
    const int H = (int)shapes[1][1], W = (int)shapes[1][2], Ctot = (int)shapes[1][3];
    if (shapes[1][0] != N || H != Hs * up || W != Ws * up || C < 8 || C % 8 || Cs % 8 || Ctot % 8 || at->c0 % 8 || at->c0 < 0 || sc0 % 8 || sc0 < 0 ||
        at->width != C || at->c0 + C > Ctot || sc0 + C > Cs)
        return MD_ERR_ARG;
    const size_t total = (size_t)N * H * W * (C / 8);
    if (total == 0) return MD_OK;
    hipLaunchKernelGGL(slice_write_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (uint16_t *)params[1], N, H, W, C, Ctot, at->c0, up, Cs, sc0);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}
extern "C" int md_concat_copy(MD_AOT_ARGS) { return slice_write_impl(nparam, params, ndims, shapes, dtypes, stream, extra, 1); }
extern "C" int md_upsample2x(MD_AOT_ARGS) { return slice_write_impl(nparam, params, ndims, shapes, dtypes, stream, extra, 2); }

extern "C" int md_slice_cast(MD_AOT_ARGS) {
    if (nparam != 2) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] < 1 || ndims[1] < 1) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32")) return MD_ERR_ARG;
    const md_slice_attrs *at = (const md_slice_attrs *)extra;
    const int C = (int)shapes[0][ndims[0] - 1];
    const int64_t tot = numel(ndims, shapes, 0);
    if (C <= 0 || at->c0 < 0 || at->width < 1 || at->c0 + at->width > C) return MD_ERR_ARG;
    const size_t M = (size_t)(tot / C);
    if (numel(ndims, shapes, 1) != (int64_t)(M * at->width)) return MD_ERR_ARG;
    if (M == 0) return MD_OK;
    hipLaunchKernelGGL(slice_cast_kernel, dim3(grid_for(M * at->width)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (float *)params[1], M, C, at->c0, at->width);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}
