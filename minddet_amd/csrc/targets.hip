// targets.hip -- anchor target assignment on the device (consumer of the a6 / a7 kernels).
//
// What it replaces: create_target_np (minddet/models/pointpillars/src/core/target_assigner.py:29-166) as
// TargetAssigner.assign drives it (:196-224) with positive_fraction None (configs/car_xyres16.yaml:129): similarity =
// iou_jit(rbbox2d_to_near_bbox(.), eps 0) (region_similarity.py:46-59, box_np_ops.py:180-192,639-679), box encoding =
// second_box_encode (box_np_ops.py:8-37).  The [A, G] similarity matrix is never materialised: pass 1 reduces the
// per-ground-truth maxima (LDS + global atomicMax on the IoU bits: IoU >= 0 orders like an unsigned integer), pass 2
// recomputes each anchor's row, takes the FIRST arg-max as numpy does, and applies the label rules in the reference's
// order (forced matches survive the background rule).  Same float32 operation order as the numpy code
// (fp contraction off), so labels / ids are bit-exact; log() in the size targets may differ in the last ulps.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

#pragma clang fp contract(off)

namespace md {

constexpr int TG_MAX_GT = 1024;

__device__ __forceinline__ float4 near_bbox(const float *b7) {
    // rbbox2d_to_near_bbox on (x, y, w, l, r) = b7[0,1,3,4,6]
    const float PI_F = 3.14159274101257324f, PI4_F = 0.785398185253143311f;
    const float r = b7[6];
    const float lp = fabsf(r - floorf(r / PI_F + 0.5f) * PI_F);
    const bool swap = lp > PI4_F;
    const float dx = swap ? b7[4] : b7[3], dy = swap ? b7[3] : b7[4];
    return make_float4(b7[0] - dx / 2, b7[1] - dy / 2, b7[0] + dx / 2, b7[1] + dy / 2);
}

__device__ __forceinline__ float iou_eps0(const float4 b, const float4 c) {  // iou_jit, eps = 0
    const float qa = (c.z - c.x) * (c.w - c.y);
    const float iw = fminf(b.z, c.z) - fmaxf(b.x, c.x);
    if (iw > 0) {
        const float ih = fminf(b.w, c.w) - fmaxf(b.y, c.y);
        if (ih > 0) {
            const float ua = (b.z - b.x) * (b.w - b.y) + qa - iw * ih;
            return iw * ih / ua;
        }
    }
    return 0.f;
}

__global__ __launch_bounds__(256) void target_colmax_kernel(const float *__restrict__ anchors, int A, const float *__restrict__ gt, int G,
                                                            const uint8_t *__restrict__ mask, unsigned *__restrict__ colmax) {
    __shared__ float4 gbox[TG_MAX_GT];
    __shared__ unsigned lmax[TG_MAX_GT];
    for (int j = threadIdx.x; j < G; j += 256) { gbox[j] = near_bbox(gt + (size_t)j * 7); lmax[j] = 0u; }
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a < A && (!mask || mask[a])) {
        const float4 ab = near_bbox(anchors + (size_t)a * 7);
        for (int j = 0; j < G; ++j) {
            const float v = iou_eps0(ab, gbox[j]);
            if (v > 0.f) atomicMax(&lmax[j], __float_as_uint(v));
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < G; j += 256)
        if (lmax[j]) atomicMax(&colmax[j], lmax[j]);
}

__global__ __launch_bounds__(256) void target_assign_kernel(const float *__restrict__ anchors, int A, const float *__restrict__ gt, int G,
                                                            const int *__restrict__ gt_cls, const float *__restrict__ mt,
                                                            const float *__restrict__ ut, const uint8_t *__restrict__ mask,
                                                            const unsigned *__restrict__ colmax, int *__restrict__ labels,
                                                            float *__restrict__ targets, float *__restrict__ weights,
                                                            int *__restrict__ gt_ids) {
    __shared__ float4 gbox[TG_MAX_GT];
    __shared__ float gmax[TG_MAX_GT];
    for (int j = threadIdx.x; j < G; j += 256) {
        gbox[j] = near_bbox(gt + (size_t)j * 7);
        const float m = __uint_as_float(colmax[j]);
        gmax[j] = m == 0.f ? -1.f : m;  // empty_gt_mask: a ground truth nothing overlaps forces no anchor
    }
    __syncthreads();
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    int label = -1, gid = -1;
    float t[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!mask || mask[a]) {
        if (G > 0) {
            const float *an = anchors + (size_t)a * 7;
            const float4 ab = near_bbox(an);
            int arg = 0;
            float amax = -1.f;
            bool forced = false;
            for (int j = 0; j < G; ++j) {
                const float v = iou_eps0(ab, gbox[j]);
                if (v > amax) { amax = v; arg = j; }  // first maximum
                forced |= v == gmax[j];
            }
            const bool fg = forced || amax >= mt[a];
            if (fg) label = gt_cls[arg];
            else if (amax < ut[a]) label = 0;
            if (label > 0) {
                gid = arg;
                const float *g = gt + (size_t)arg * 7;  // second_box_encode(gt[arg], anchor)
                const float zg = g[2] + g[5] / 2, za = an[2] + an[5] / 2;
                const float diagonal = sqrtf(an[4] * an[4] + an[3] * an[3]);
                t[0] = (g[0] - an[0]) / diagonal;
                t[1] = (g[1] - an[1]) / diagonal;
                t[2] = (zg - za) / an[5];
                t[3] = logf(g[3] / an[3]);
                t[4] = logf(g[4] / an[4]);
                t[5] = logf(g[5] / an[5]);
                t[6] = g[6] - an[6];
            }
        } else {
            label = 0;
        }
    }
    labels[a] = label;
    gt_ids[a] = gid;
    weights[a] = label > 0 ? 1.f : 0.f;
#pragma unroll
    for (int i = 0; i < 7; ++i) targets[(size_t)a * 7 + i] = t[i];
}

}  // namespace md

using namespace md;

extern "C" int md_assign_targets(MD_AOT_ARGS) {
    // in : anchors[A,7] f32, gt[G,7] f32, gt_cls[G] i32, matched_thr[A] f32, unmatched_thr[A] f32, mask[A] u8 | NULL
    // out: labels[A] i32, bbox_targets[A,7] f32, bbox_outside_weights[A] f32, gt_ids[A] i32 ; [workspace >= 4*G bytes]
    if (nparam != 10 && nparam != 11) return MD_ERR_NPARAM;
    if (!params || !ndims || !shapes) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 3, "float32") || !dtype_is(dtypes, 4, "float32") ||
        !dtype_is(dtypes, 6, "int32") || !dtype_is(dtypes, 7, "float32") || !dtype_is(dtypes, 8, "float32") || !dtype_is(dtypes, 9, "int32"))
        return MD_ERR_ARG;
    const int64_t A = dim(ndims, shapes, 0, 0), G = params[1] ? dim(ndims, shapes, 1, 0) : 0;
    if (A < 0 || G < 0 || dim(ndims, shapes, 0, 1) != 7 || (G > 0 && dim(ndims, shapes, 1, 1) != 7)) return MD_ERR_ARG;
    if (numel(ndims, shapes, 3) != A || numel(ndims, shapes, 4) != A || numel(ndims, shapes, 6) != A || numel(ndims, shapes, 7) != A * 7 ||
        numel(ndims, shapes, 8) != A || numel(ndims, shapes, 9) != A || (G > 0 && numel(ndims, shapes, 2) != G))
        return MD_ERR_ARG;
    if (params[5] && (!dtype_is(dtypes, 5, "uint8") || numel(ndims, shapes, 5) != A)) return MD_ERR_ARG;
    if (G > 0 && (!dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int32"))) return MD_ERR_ARG;
    if (A == 0) return MD_OK;
    if (A > 0x7fffff00LL || G > TG_MAX_GT) return MD_ERR_SIZE;
    if (!params[0] || !params[3] || !params[4] || !params[6] || !params[7] || !params[8] || !params[9] || (G > 0 && (!params[1] || !params[2])))
        return MD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    Scratch ws;
    const int rc = ws.acquire((size_t)(G > 0 ? G : 1) * 4, nparam, params, ndims, shapes, 10, s);
    if (rc != MD_OK) return rc;
    unsigned *colmax = (unsigned *)ws.ptr;
    const unsigned blocks = (unsigned)((A + 255) / 256);
    if (G > 0) {
        MD_HIP_TRY(hipMemsetAsync(colmax, 0, (size_t)G * 4, s));
        hipLaunchKernelGGL(target_colmax_kernel, dim3(blocks), dim3(256), 0, s, (const float *)params[0], (int)A, (const float *)params[1],
                           (int)G, (const uint8_t *)params[5], colmax);
    }
    hipLaunchKernelGGL(target_assign_kernel, dim3(blocks), dim3(256), 0, s, (const float *)params[0], (int)A, (const float *)params[1], (int)G,
                       (const int *)params[2], (const float *)params[3], (const float *)params[4], (const uint8_t *)params[5], colmax,
                       (int *)params[6], (float *)params[7], (float *)params[8], (int *)params[9]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}
