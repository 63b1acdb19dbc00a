// twostage.hip -- the small glue kernels of the two-stage (Faster R-CNN) inference path between the
// conv stacks and the detection ops: RPN proposal decode, per-image merge across FPN levels, RoI
// list assembly, class-wise candidate scoring, selected-candidate decode and detection packing.
//
// None of this exists in the reference (Faster R-CNN is a README bullet); the contract
// follows the public mmdet/torchvision definitions and is pinned only by the build's own oracle
// (oracle/pipeline.py) -- "parity unpinned".  The per-class NMS merge + top-100 packing mirrors the
// SHAPE of minddet/models/centernet/src/post_process.py:36-61 (per-class loop, global top-k by score).
//
// Everything stays on device, fixed-shape and padded, so the whole step is stream-ordered with
// no host round trip (the reference's eval loops go device->host->device per image:
// minddet/models/pointpillars/src/predict.py:273-328).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "aot.h"

namespace md {

__device__ __forceinline__ float tbf2f(unsigned v16) { return __uint_as_float(v16 << 16); }

static inline unsigned tgrid(size_t total) {
    size_t b = (total + 255) / 256;
    return (unsigned)(b > 16384 ? 16384 : (b == 0 ? 1 : b));
}

struct DecodeP { float mean[4], stdv[4]; float max_ratio, clip_w, clip_h; };

__device__ __forceinline__ float4 decode_box(float4 r, float dx, float dy, float dw, float dh, const DecodeP &p) {
    dx = dx * p.stdv[0] + p.mean[0]; dy = dy * p.stdv[1] + p.mean[1];
    dw = dw * p.stdv[2] + p.mean[2]; dh = dh * p.stdv[3] + p.mean[3];
    dw = fminf(fmaxf(dw, -p.max_ratio), p.max_ratio);
    dh = fminf(fmaxf(dh, -p.max_ratio), p.max_ratio);
    const float px = (r.x + r.z) * 0.5f, py = (r.y + r.w) * 0.5f, pw = r.z - r.x, ph = r.w - r.y;
    const float gw = pw * expf(dw), gh = ph * expf(dh);
    const float gx = px + pw * dx, gy = py + ph * dy;
    float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
    if (p.clip_w > 0.f) {
        x1 = fminf(fmaxf(x1, 0.f), p.clip_w); x2 = fminf(fmaxf(x2, 0.f), p.clip_w);
        y1 = fminf(fmaxf(y1, 0.f), p.clip_h); y2 = fminf(fmaxf(y2, 0.f), p.clip_h);
    }
    return make_float4(x1, y1, x2, y2);
}

// head [B, HW, Cp] bf16: channels [0,A) objectness logits, [A, 5A) deltas (anchor-major: a*4+j).
// idx[B,k] indexes (loc*A + a) within the level.
__global__ void rpn_decode_kernel(const uint16_t *__restrict__ head, const float *__restrict__ anchors,
                                  const int *__restrict__ idx, const int *__restrict__ cnt, int B, int k, int HW, int A,
                                  int Cp, DecodeP p, float *__restrict__ boxes, float *__restrict__ scores) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * k) return;
    const int b = e / k, j = e % k;
    float4 o = make_float4(0, 0, 0, 0);
    float s = -FLT_MAX;
    if (j < cnt[b]) {
        const int id = idx[e];
        const int loc = id / A, a = id % A;
        const uint16_t *h = head + ((size_t)b * HW + loc) * Cp;
        const float logit = tbf2f(h[a]);
        const float4 an = *reinterpret_cast<const float4 *>(anchors + (size_t)id * 4);
        o = decode_box(an, tbf2f(h[A + a * 4]), tbf2f(h[A + a * 4 + 1]), tbf2f(h[A + a * 4 + 2]), tbf2f(h[A + a * 4 + 3]), p);
        s = 1.0f / (1.0f + expf(-logit));
    }
    *reinterpret_cast<float4 *>(boxes + (size_t)e * 4) = o;
    scores[e] = s;
}

// [L,B,k] level-major lists -> per-image [B, L*k]; suppressed / invalid entries get score -FLT_MAX
__global__ void rpn_merge_kernel(const float *__restrict__ boxes, const float *__restrict__ scores,
                                 const unsigned char *__restrict__ keep, int L, int B, int k,
                                 float *__restrict__ mboxes, float *__restrict__ mscores) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= L * B * k) return;
    const int l = e / (B * k), r = e % (B * k), b = r / k, j = r % k;
    const size_t dst = ((size_t)b * L + l) * k + j;
    *reinterpret_cast<float4 *>(mboxes + dst * 4) = *reinterpret_cast<const float4 *>(boxes + (size_t)e * 4);
    mscores[dst] = keep[e] ? scores[e] : -FLT_MAX;
}

// rois [B*post, 5] = (b, box) ; invalid slots (j >= cnt[b]) become a zero box with batch index b
__global__ void make_rois_kernel(const float *__restrict__ mboxes, const float *__restrict__ topv,
                                 const int *__restrict__ topi, const int *__restrict__ cnt, int B, int post, int per_img,
                                 float *__restrict__ rois, float *__restrict__ roi_scores) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * post) return;
    const int b = e / post, j = e % post;
    float4 bx = make_float4(0, 0, 0, 0);
    float s = 0.f;
    if (j < cnt[b]) {
        bx = *reinterpret_cast<const float4 *>(mboxes + ((size_t)b * per_img + topi[e]) * 4);
        s = topv[e];
    }
    float *o = rois + (size_t)e * 5;
    o[0] = (float)b; o[1] = bx.x; o[2] = bx.y; o[3] = bx.z; o[4] = bx.w;
    roi_scores[e] = s;
}

// cls_reg [R, Cp] bf16: channels [0, nc+1) class logits (background LAST, index nc),
// [reg0, reg0 + 4*nc) class-specific deltas.  One wave per RoI: softmax over nc+1 logits with a
// cross-lane max/sum, then cand[b, j*nc + c] = p_c if p_c > thr (and the RoI slot is valid) else -FLT_MAX.
__global__ __launch_bounds__(256) void rcnn_scores_kernel(const uint16_t *__restrict__ cls_reg,
                                                           const int *__restrict__ roi_cnt, int R, int post, int nc,
                                                           int Cp, float thr, float *__restrict__ cand) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const int b = r / post, j = r % post;
    const bool valid = j < roi_cnt[b];
    const uint16_t *h = cls_reg + (size_t)r * Cp;
    float v[2];
    float m = -FLT_MAX;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = lane + 64 * t;
        v[t] = c <= nc ? tbf2f(h[c]) : -FLT_MAX;
        m = fmaxf(m, v[t]);
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    float ssum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = lane + 64 * t;
        v[t] = c <= nc ? expf(v[t] - m) : 0.f;
        ssum += v[t];
    }
    for (int o = 32; o > 0; o >>= 1) ssum += __shfl_xor(ssum, o, 64);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int c = lane + 64 * t;
        if (c < nc) {
            const float p = v[t] / ssum;
            cand[(size_t)r * nc + c] = (valid && p > thr) ? p : -FLT_MAX;
        }
    }
}

// selected candidates -> boxes/labels.  sel_idx[B,npre] indexes (j*nc + c) within image b.
__global__ void rcnn_decode_selected_kernel(const uint16_t *__restrict__ cls_reg, const float *__restrict__ rois,
                                            const int *__restrict__ sel_idx, const int *__restrict__ sel_cnt, int B,
                                            int npre, int post, int nc, int Cp, int reg0, DecodeP p,
                                            float *__restrict__ boxes, int *__restrict__ labels) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * npre) return;
    const int b = e / npre, q = e % npre;
    float4 o = make_float4(0, 0, 0, 0);
    int lab = -1;
    if (q < sel_cnt[b]) {
        const int id = sel_idx[e];
        const int j = id / nc, c = id % nc;
        const int r = b * post + j;
        const float *roi = rois + (size_t)r * 5;
        const uint16_t *h = cls_reg + (size_t)r * Cp + reg0 + c * 4;
        o = decode_box(make_float4(roi[1], roi[2], roi[3], roi[4]), tbf2f(h[0]), tbf2f(h[1]), tbf2f(h[2]), tbf2f(h[3]), p);
        lab = c;
    }
    *reinterpret_cast<float4 *>(boxes + (size_t)e * 4) = o;
    labels[e] = lab;
}

// dets[B, max_det, 6] = x1,y1,x2,y2,score,label for the first num[b] survivors (score order), rest 0; count[B]
__global__ void pack_dets_kernel(const float *__restrict__ boxes, const float *__restrict__ scores,
                                 const int *__restrict__ labels, const int *__restrict__ keep_idx,
                                 const int *__restrict__ num, int B, int npre, int max_det, float *__restrict__ dets,
                                 int *__restrict__ count, const int *__restrict__ sel_cnt, int *__restrict__ status) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * max_det) return;
    const int b = e / max_det, j = e % max_det;
    const int n = min(num[b], max_det);
    float *d = dets + (size_t)e * 6;
    if (j < n) {
        const int q = keep_idx[(size_t)b * npre + j];
        const float4 bx = *reinterpret_cast<const float4 *>(boxes + ((size_t)b * npre + q) * 4);
        d[0] = bx.x; d[1] = bx.y; d[2] = bx.z; d[3] = bx.w;
        d[4] = scores[(size_t)b * npre + q];
        d[5] = (float)labels[(size_t)b * npre + q];
    } else {
        d[0] = d[1] = d[2] = d[3] = d[4] = d[5] = 0.f;
    }
    if (j == 0) {
        count[b] = n;
        // the greedy class-wise NMS ran on the top-npre candidates only.  Its first max_det survivors ARE the first max_det
        // survivors of the untruncated list (a candidate is suppressed by higher-scored ones only) unless the prefix was full
        // and ran out before the quota: then candidates beyond it may belong to the result.  Bit 0 is OR-ed in (the caller
        // owns / clears the word), so a flag raised in any step stays visible without a host synchronisation per step.
        if (status && sel_cnt && num[b] < max_det && sel_cnt[b] >= npre) status[b] |= 1;
    }
}

static void fill_decode(DecodeP &p, const md_delta2bbox_attrs &a) {
    for (int i = 0; i < 4; ++i) { p.mean[i] = a.means[i]; p.stdv[i] = a.stds[i]; }
    p.max_ratio = a.max_ratio;
    p.clip_w = (a.clip_w > 0 && a.clip_h > 0) ? a.clip_w : 0.f;
    p.clip_h = a.clip_h;
}

}  // namespace md

using namespace md;

extern "C" int md_rpn_decode(MD_AOT_ARGS) {
    // in: head[B,H,W,Cp] bf16, anchors[HWA,4] f32, idx[B,k] i32, cnt[B] i32 ; out: boxes[B,k,4] f32, scores[B,k] f32
    if (nparam != 6) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || ndims[0] != 4 || ndims[2] != 2) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "int32") || !dtype_is(dtypes, 4, "float32") || !dtype_is(dtypes, 5, "float32"))
        return MD_ERR_ARG;
    const md_rpn_decode_attrs *at = (const md_rpn_decode_attrs *)extra;
    const int B = (int)shapes[0][0], HW = (int)(shapes[0][1] * shapes[0][2]), Cp = (int)shapes[0][3];
    const int k = (int)shapes[2][1], A = at->num_anchors;
    if (shapes[2][0] != B || A < 1 || 5 * A > Cp || numel(ndims, shapes, 1) != (int64_t)HW * A * 4) return MD_ERR_ARG;
    if (numel(ndims, shapes, 4) != (int64_t)B * k * 4 || numel(ndims, shapes, 5) != (int64_t)B * k) return MD_ERR_ARG;
    if (B * k == 0) return MD_OK;
    DecodeP p;
    fill_decode(p, at->decode);
    hipLaunchKernelGGL(rpn_decode_kernel, dim3((B * k + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (const float *)params[1], (const int *)params[2], (const int *)params[3],
                       B, k, HW, A, Cp, p, (float *)params[4], (float *)params[5]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_rpn_merge(MD_AOT_ARGS) {
    // in: boxes[L,B,k,4] f32, scores[L,B,k] f32, keep[L,B,k] u8 ; out: mboxes[B,L*k,4] f32, mscores[B,L*k] f32
    if (nparam != 5) return MD_ERR_NPARAM;
    if (!params || !ndims || ndims[1] != 3) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "uint8") ||
        !dtype_is(dtypes, 3, "float32") || !dtype_is(dtypes, 4, "float32"))
        return MD_ERR_ARG;
    const int L = (int)shapes[1][0], B = (int)shapes[1][1], k = (int)shapes[1][2];
    const int64_t tot = (int64_t)L * B * k;
    if (numel(ndims, shapes, 0) != tot * 4 || numel(ndims, shapes, 2) != tot || numel(ndims, shapes, 3) != tot * 4 ||
        numel(ndims, shapes, 4) != tot)
        return MD_ERR_ARG;
    if (tot == 0) return MD_OK;
    hipLaunchKernelGGL(rpn_merge_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (const float *)params[1], (const unsigned char *)params[2], L, B, k,
                       (float *)params[3], (float *)params[4]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_make_rois(MD_AOT_ARGS) {
    // in: mboxes[B,P,4] f32, topv[B,post] f32, topi[B,post] i32, cnt[B] i32 ; out: rois[B*post,5] f32, scores[B*post] f32
    if (nparam != 6) return MD_ERR_NPARAM;
    if (!params || !ndims || ndims[0] != 3 || ndims[1] != 2) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "int32") || !dtype_is(dtypes, 4, "float32") || !dtype_is(dtypes, 5, "float32"))
        return MD_ERR_ARG;
    const int B = (int)shapes[0][0], per = (int)shapes[0][1], post = (int)shapes[1][1];
    if (shapes[1][0] != B || numel(ndims, shapes, 4) != (int64_t)B * post * 5 || numel(ndims, shapes, 5) != (int64_t)B * post)
        return MD_ERR_ARG;
    if (B * post == 0) return MD_OK;
    hipLaunchKernelGGL(make_rois_kernel, dim3((B * post + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (const float *)params[1], (const int *)params[2], (const int *)params[3],
                       B, post, per, (float *)params[4], (float *)params[5]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_rcnn_scores(MD_AOT_ARGS) {
    // in: cls_reg[R,Cp] bf16, roi_cnt[B] i32 ; out: cand[B, post*nc] f32
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || ndims[0] != 2) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "int32") || !dtype_is(dtypes, 2, "float32")) return MD_ERR_ARG;
    const md_rcnn_attrs *at = (const md_rcnn_attrs *)extra;
    const int R = (int)shapes[0][0], Cp = (int)shapes[0][1];
    const int B = (int)numel(ndims, shapes, 1);
    if (B < 1 || R % B || at->num_classes < 1 || at->num_classes > 127 || at->num_classes + 1 > Cp) return MD_ERR_ARG;
    if (numel(ndims, shapes, 2) != (int64_t)R * at->num_classes) return MD_ERR_ARG;
    if (R == 0) return MD_OK;
    hipLaunchKernelGGL(rcnn_scores_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (const int *)params[1], R, R / B, at->num_classes, Cp, at->score_thr,
                       (float *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_rcnn_decode_selected(MD_AOT_ARGS) {
    // in: cls_reg[R,Cp] bf16, rois[R,5] f32, sel_idx[B,npre] i32, sel_cnt[B] i32 ; out: boxes[B,npre,4] f32, labels[B,npre] i32
    if (nparam != 6) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || ndims[0] != 2 || ndims[2] != 2) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "int32") || !dtype_is(dtypes, 4, "float32") || !dtype_is(dtypes, 5, "int32"))
        return MD_ERR_ARG;
    const md_rcnn_attrs *at = (const md_rcnn_attrs *)extra;
    const int R = (int)shapes[0][0], Cp = (int)shapes[0][1], B = (int)shapes[2][0], npre = (int)shapes[2][1];
    if (B < 1 || R % B || at->reg_offset + 4 * at->num_classes > Cp) return MD_ERR_ARG;
    if (numel(ndims, shapes, 4) != (int64_t)B * npre * 4 || numel(ndims, shapes, 5) != (int64_t)B * npre) return MD_ERR_ARG;
    if (B * npre == 0) return MD_OK;
    DecodeP p;
    fill_decode(p, at->decode);
    hipLaunchKernelGGL(rcnn_decode_selected_kernel, dim3((B * npre + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (const float *)params[1], (const int *)params[2], (const int *)params[3],
                       B, npre, R / B, at->num_classes, Cp, at->reg_offset, p, (float *)params[4], (int *)params[5]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_pack_detections(MD_AOT_ARGS) {
    // in: boxes[B,npre,4] f32, scores[B,npre] f32, labels[B,npre] i32, keep_idx[B,npre] i32, num[B] i32 [, sel_cnt[B] i32]
    // out: dets[B,max_det,6] f32, count[B] i32 [, status[B] i32 (in/out: bit 0 OR-ed in)]
    if (nparam != 7 && nparam != 9) return MD_ERR_NPARAM;
    const int o = nparam == 9 ? 6 : 5;   // index of dets
    if (!params || !ndims || ndims[1] != 2 || ndims[o] != 3) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "int32") || !dtype_is(dtypes, 4, "int32") || !dtype_is(dtypes, o, "float32") ||
        !dtype_is(dtypes, o + 1, "int32"))
        return MD_ERR_ARG;
    const int B = (int)shapes[1][0], npre = (int)shapes[1][1], max_det = (int)shapes[o][1];
    if (shapes[o][0] != B || shapes[o][2] != 6) return MD_ERR_ARG;
    const int *sel_cnt = nullptr;
    int *status = nullptr;
    if (nparam == 9) {
        if (!dtype_is(dtypes, 5, "int32") || !dtype_is(dtypes, 8, "int32") || numel(ndims, shapes, 5) != B || numel(ndims, shapes, 8) != B ||
            (B > 0 && (!params[5] || !params[8])))
            return MD_ERR_ARG;
        sel_cnt = (const int *)params[5]; status = (int *)params[8];
    }
    if (B * max_det == 0) return MD_OK;
    hipLaunchKernelGGL(pack_dets_kernel, dim3((B * max_det + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (const float *)params[1], (const int *)params[2], (const int *)params[3],
                       (const int *)params[4], B, npre, max_det, (float *)params[o], (int *)params[o + 1], sel_cnt, status);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

// ------------------------------------------------------------------------------------------ Mask R-CNN mask select
// logits [R,S,S,Cpad] bf16 (per-class mask logits), dets [R,6] f32 (x1,y1,x2,y2,score,label) -> masks [R,S,S] f32 =
// sigmoid(logit of the detection's own class); rows of empty detection slots (score 0) give zeros.  Absent from the
// reference (Mask R-CNN is a README bullet): standard head, parity unpinned.
namespace md {
__global__ void mask_select_kernel(const uint16_t *__restrict__ logits, const float *__restrict__ dets, int R, int SS, int C, int nc,
                                   float *__restrict__ out) {
    const size_t total = (size_t)R * SS;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e / SS);
        const float score = dets[(size_t)r * 6 + 4];
        const int label = (int)dets[(size_t)r * 6 + 5];
        float v = 0.f;
        if (score > 0.f && label >= 0 && label < nc) {
            const float x = __uint_as_float((unsigned)logits[e * C + label] << 16);
            v = 1.0f / (1.0f + expf(-x));
        }
        out[e] = v;
    }
}
}  // namespace md

extern "C" int md_mask_select(MD_AOT_ARGS) {
    // in: logits[R,S,S,Cpad] bf16, dets[R,6] f32 ; out: masks[R,S,S] f32 ; extra: int32 num_classes
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4) return MD_ERR_ARG;
    if (!md::dtype_is(dtypes, 0, "bfloat16") || !md::dtype_is(dtypes, 1, "float32") || !md::dtype_is(dtypes, 2, "float32")) return MD_ERR_ARG;
    const int64_t R = shapes[0][0], S = shapes[0][1], C = shapes[0][3];
    const int nc = *(const int32_t *)extra;
    if (shapes[0][2] != S || nc < 1 || nc > C || md::numel(ndims, shapes, 1) != R * 6 || md::numel(ndims, shapes, 2) != R * S * S) return MD_ERR_ARG;
    if (R * S * S == 0) return MD_OK;
    if (!params[0] || !params[1] || !params[2]) return MD_ERR_ARG;
    const size_t total = (size_t)R * S * S;
    const size_t nb = (total + 255) / 256;
    hipLaunchKernelGGL(md::mask_select_kernel, dim3((unsigned)(nb < 65535u * 64 ? nb : 65535u * 64)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], (const float *)params[1], (int)R, (int)(S * S), (int)C, nc, (float *)params[2]);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}


// ------------------------------------------------------------------------------------------------ Mask R-CNN mask pasting
// The last step of the mask branch (BASELINE configs[4]): each detection's S x S mask probabilities are resampled bilinearly onto
// the pixels of its box at IMAGE resolution and thresholded -- the public definition (mmdet _do_paste_mask / torchvision
// paste_masks_in_image: grid_sample(align_corners=False, zero padding) at pixel centres, mask >= threshold).  Absent from the
// reference (Mask R-CNN is a README bullet, /root/reference/README.md:5-14): parity unpinned; oracle/np_ops.py::paste_masks restates
// the SAME fp32 operation sequence (no fused multiply-adds: __fmul_rn / __fadd_rn below), so the comparison is bit-exact.
// HBM-bound write kernel: one lane = one 32-pixel word of the bit mask (64 lanes = 256 contiguous bytes) or 4 pixels of the uint8
// mask; words outside the box's support are written as zeros without touching the mask.
//   u  = ((px + 0.5) - x0) * (1 / (x1 - x0));  ix = u * S - 0.5;  xl = floor(ix), fx = ix - xl  (same for y)
//   v  = ((m[yl][xl] * (1 - fx) + m[yl][xl+1] * fx) * (1 - fy)) + ((m[yl+1][xl] * (1 - fx) + m[yl+1][xl+1] * fx) * fy), taps outside = 0
#pragma clang fp contract(off)   // from here to the end of the file: every product and sum below rounds on its own, as in the numpy oracle
namespace md {
struct PasteArgs {
    const float *masks;   // [R,S,S]
    const float *dets;    // [R,6]
    void *out;            // BITS: uint32 [R,H,Ww]; else uint8 [R,H,W]
    int R, S, H, W, Ww;   // Ww = words (BITS) or 4-pixel groups (uint8) per row
    float thr;
};

template <bool BITS>
__global__ __launch_bounds__(256) void paste_masks_kernel(PasteArgs a) {
    constexpr int PX = BITS ? 32 : 4;
    // 32-bit index arithmetic (the host checks R * H * Ww + one grid stride < 2^32): 64-bit divisions were the larger part of this write-bound kernel's time
    const unsigned total = (unsigned)a.R * (unsigned)a.H * (unsigned)a.Ww;
    for (unsigned e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const unsigned t = e / (unsigned)a.Ww;
        const int wx = (int)(e - t * (unsigned)a.Ww);
        const int r = (int)(t / (unsigned)a.H), py = (int)(t - (unsigned)r * (unsigned)a.H);
        const float *d = a.dets + (size_t)r * 6;
        const float x0 = d[0], y0 = d[1], bw = __fsub_rn(d[2], d[0]), bh = __fsub_rn(d[3], d[1]);
        unsigned word = 0u;
        if (d[4] > 0.f && bw > 0.f && bh > 0.f) {
            const float Sf = (float)a.S;
            const float inv_w = __fdiv_rn(1.0f, bw), inv_h = __fdiv_rn(1.0f, bh);
            const float iy = __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fadd_rn((float)py, 0.5f), y0), inv_h), Sf), 0.5f);
            const int px0 = wx * PX;
            const float ix_first = __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fadd_rn((float)px0, 0.5f), x0), inv_w), Sf), 0.5f);
            const float ix_last = __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fadd_rn((float)(px0 + PX - 1), 0.5f), x0), inv_w), Sf), 0.5f);
            if (iy > -1.0f && iy < Sf && ix_last > -1.0f && ix_first < Sf) {
                const float yl_f = floorf(iy);
                const int yl = (int)yl_f;
                const float fy = __fsub_rn(iy, yl_f), gy = __fsub_rn(1.0f, fy);
                const float *m0 = a.masks + (size_t)r * a.S * a.S + (size_t)(yl < 0 ? 0 : yl) * a.S;
                const float *m1 = a.masks + (size_t)r * a.S * a.S + (size_t)(yl + 1 < a.S ? yl + 1 : a.S - 1) * a.S;
                const bool ok0 = yl >= 0, ok1 = yl + 1 < a.S;
#pragma unroll 4
                for (int j = 0; j < PX; ++j) {
                    const int px = px0 + j;
                    const float ix = __fsub_rn(__fmul_rn(__fmul_rn(__fsub_rn(__fadd_rn((float)px, 0.5f), x0), inv_w), Sf), 0.5f);
                    if (!(ix > -1.0f && ix < Sf) || px >= a.W) continue;
                    const float xl_f = floorf(ix);
                    const int xl = (int)xl_f;
                    const float fx = __fsub_rn(ix, xl_f), gx = __fsub_rn(1.0f, fx);
                    const bool okl = xl >= 0, okr = xl + 1 < a.S;
                    const int cl = okl ? xl : 0, cr = okr ? xl + 1 : a.S - 1;
                    const float v00 = ok0 && okl ? m0[cl] : 0.f, v01 = ok0 && okr ? m0[cr] : 0.f;
                    const float v10 = ok1 && okl ? m1[cl] : 0.f, v11 = ok1 && okr ? m1[cr] : 0.f;
                    const float top = __fadd_rn(__fmul_rn(v00, gx), __fmul_rn(v01, fx));
                    const float bot = __fadd_rn(__fmul_rn(v10, gx), __fmul_rn(v11, fx));
                    const float v = __fadd_rn(__fmul_rn(top, gy), __fmul_rn(bot, fy));
                    if (v >= a.thr) word |= BITS ? (1u << j) : (1u << (8 * j));
                }
            }
        }
        if (BITS) {
            reinterpret_cast<unsigned *>(a.out)[e] = word;
        } else {   // 4 pixels of the uint8 mask; the last group of a ragged row is stored byte by byte
            unsigned char *o = reinterpret_cast<unsigned char *>(a.out) + ((size_t)r * a.H + py) * a.W + (size_t)wx * 4;
            if (wx * 4 + 4 <= a.W && (a.W & 3) == 0) *reinterpret_cast<unsigned *>(o) = word;
            else
                for (int j = 0; j < 4 && wx * 4 + j < a.W; ++j) o[j] = (unsigned char)((word >> (8 * j)) & 1u);
        }
    }
}
}  // namespace md

extern "C" int md_paste_masks(MD_AOT_ARGS) {
    // in: masks[R,S,S] f32, dets[R,6] f32 ; out: bits != 0: words[R,H,ceil(W/32)] i32 (bit j of word k = pixel 32 k + j), else mask[R,H,W] u8
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 3 || ndims[2] != 3) return MD_ERR_ARG;
    const md_paste_attrs *at = (const md_paste_attrs *)extra;
    if (!md::dtype_is(dtypes, 0, "float32") || !md::dtype_is(dtypes, 1, "float32") || !md::dtype_is(dtypes, 2, at->bits ? "int32" : "uint8"))
        return MD_ERR_ARG;
    const int64_t R = shapes[0][0], S = shapes[0][1], H = at->img_h, W = at->img_w;
    if (shapes[0][2] != S || S < 1 || S > 1024 || H < 1 || W < 1 || H > 32768 || W > 32768 || md::numel(ndims, shapes, 1) != R * 6) return MD_ERR_ARG;
    const int64_t Ww = at->bits ? (W + 31) / 32 : (W + 3) / 4;
    if (shapes[2][0] != R || shapes[2][1] != H || shapes[2][2] != (at->bits ? Ww : W)) return MD_ERR_ARG;
    if (R == 0) return MD_OK;
    if (!params[0] || !params[1] || !params[2]) return MD_ERR_ARG;
    // the kernel's grid-stride index is 32-bit: e += grid * 256 must not wrap below `total` (r03 ADVICE: it would restart the loop and never end)
    if (R > 0x7fffffffLL / 6 || (long long)R * H * Ww > 0xffffffffLL - 256LL * 64 * 256) return MD_ERR_SIZE;
    md::PasteArgs a = {(const float *)params[0], (const float *)params[1], params[2], (int)R, (int)S, (int)H, (int)W, (int)Ww, at->threshold};
    const size_t total = (size_t)R * H * Ww, nb = (total + 255) / 256;
    const unsigned grid = (unsigned)(nb < 256u * 64 ? nb : 256u * 64);   // grid-stride: 64 blocks per CU
    if (at->bits) hipLaunchKernelGGL(md::paste_masks_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(md::paste_masks_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}
