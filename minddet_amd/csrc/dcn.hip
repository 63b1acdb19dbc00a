// dcn.hip -- modulated deformable convolution (DCNv2) as deformable im2col + the MFMA GEMM.
//
// What it replaces: minddet/models/centernet/src/resnet.py:24-106 (ModulatedDeformConv2d: conv_offset -> chunk(o1, o2, mask)
// -> sigmoid(mask) -> ops.deformable_conv2d) in the CenterNet neck (centernet_det.py:123-160).  The arithmetic of
// ops.deformable_conv2d lives inside MindSpore (un-vendored): this follows the published DCNv2 definition the wrapper's
// layout implies -- for tap k = ky*3 + kx the offset conv's channels (2k, 2k+1) are (dy, dx), channel 18 + k is the mask
// logit; y = sum_k sigmoid(mask_k) * W_k . bilinear(x, p0 + p_k + (dy_k, dx_k)), samples outside the image read as zero.
// Parity unpinned.
//
// md_deform_cols writes the modulated, bilinearly sampled columns [N,Ho,Wo, kh*kw*C] bf16 (K ordered (tap, channel), which
// is exactly the K order of the packed conv weights), then md_conv2d runs the product as a 1x1 conv on the MFMA kernels.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

namespace md {

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float dbf2f(uint16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ unsigned dpk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

struct DcnArgs {
    const uint16_t *x;    // [N,H,W,C]
    const uint16_t *off;  // [N,Ho,Wo,Coff]: 2*T offset channels (dy, dx per tap), then T mask logits
    uint16_t *cols;       // [N,Ho,Wo,T*C]
    int N, H, W, C, Ho, Wo, Coff, kh, kw, stride, pad;
};

// one lane = 8 channels of one (output pixel, tap); consecutive lanes = consecutive channel chunks
__global__ __launch_bounds__(256) void deform_cols_kernel(DcnArgs a, size_t total) {
    const int cv = a.C / 8, T = a.kh * a.kw;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(e % cv);
        size_t t = e / cv;
        const int k = (int)(t % T);
        const size_t pix = t / T;  // (n*Ho + ho)*Wo + wo
        const int wo = (int)(pix % a.Wo);
        const int ho = (int)((pix / a.Wo) % a.Ho);
        const int n = (int)(pix / ((size_t)a.Wo * a.Ho));
        const uint16_t *o = a.off + pix * a.Coff;
        const float dy = dbf2f(o[2 * k]), dx = dbf2f(o[2 * k + 1]);
        const float m = 1.0f / (1.0f + __expf(-dbf2f(o[2 * T + k])));
        const float y = (float)(ho * a.stride - a.pad + k / a.kw) + dy;
        const float x = (float)(wo * a.stride - a.pad + k % a.kw) + dx;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (y > -1.f && y < (float)a.H && x > -1.f && x < (float)a.W) {
            const float yf = floorf(y), xf = floorf(x);
            const int y0 = (int)yf, x0 = (int)xf;
            const float ly = y - yf, lx = x - xf, hy = 1.f - ly, hx = 1.f - lx;
            const uint16_t *base = a.x + (size_t)n * a.H * a.W * a.C + c8 * 8;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int yy = y0 + (q >> 1), xx = x0 + (q & 1);
                const float w = ((q >> 1) ? ly : hy) * ((q & 1) ? lx : hx);
                if ((unsigned)yy < (unsigned)a.H && (unsigned)xx < (unsigned)a.W) {
                    const u32x4 v = *reinterpret_cast<const u32x4 *>(base + ((size_t)yy * a.W + xx) * a.C);
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        acc[2 * d] += w * __uint_as_float(v[d] << 16);
                        acc[2 * d + 1] += w * __uint_as_float(v[d] & 0xffff0000u);
                    }
                }
            }
        }
        u32x4 outv;
#pragma unroll
        for (int d = 0; d < 4; ++d) outv[d] = dpk_bf16(acc[2 * d] * m, acc[2 * d + 1] * m);
        *reinterpret_cast<u32x4 *>(a.cols + (pix * T + k) * a.C + c8 * 8) = outv;
    }
}

}  // namespace md

using namespace md;

extern "C" int md_deform_cols(MD_AOT_ARGS) {
    // in: x[N,H,W,C] bf16, off[N,Ho,Wo,Coff >= 3*kh*kw] bf16 ; out: cols[N,Ho,Wo,kh*kw*C] bf16 ; extra: md_pool_attrs (k,stride,pad)
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4 || ndims[1] != 4 || ndims[2] != 4) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16") || !dtype_is(dtypes, 2, "bfloat16")) return MD_ERR_ARG;
    const md_pool_attrs *at = (const md_pool_attrs *)extra;
    DcnArgs a;
    a.N = (int)shapes[0][0]; a.H = (int)shapes[0][1]; a.W = (int)shapes[0][2]; a.C = (int)shapes[0][3];
    a.Ho = (int)shapes[1][1]; a.Wo = (int)shapes[1][2]; a.Coff = (int)shapes[1][3];
    a.kh = a.kw = at->k; a.stride = at->stride; a.pad = at->pad;
    if (a.kh < 1 || a.kh > 7 || a.stride < 1 || a.pad < 0 || a.C % 8) return MD_ERR_ARG;
    const int T = a.kh * a.kw;
    if (a.Coff < 3 * T || shapes[1][0] != a.N || a.Ho != (a.H + 2 * a.pad - a.kh) / a.stride + 1 ||
        a.Wo != (a.W + 2 * a.pad - a.kw) / a.stride + 1)
        return MD_ERR_ARG;
    if (shapes[2][0] != a.N || shapes[2][1] != a.Ho || shapes[2][2] != a.Wo || shapes[2][3] != (int64_t)T * a.C) return MD_ERR_ARG;
    const size_t total = (size_t)a.N * a.Ho * a.Wo * T * (a.C / 8);
    if (total == 0) return MD_OK;
    if (!params[0] || !params[1] || !params[2]) return MD_ERR_ARG;
    a.x = (const uint16_t *)params[0]; a.off = (const uint16_t *)params[1]; a.cols = (uint16_t *)params[2];
    const size_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(deform_cols_kernel, dim3((unsigned)(nb < 0x7fffffffull ? nb : 0x7fffffffull)), dim3(256), 0, (hipStream_t)stream, a, total);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}
