// nms.hip -- NMS family + IoU matrices for gfx950 (MI355X), behind the AOT C ABI.
//
// Reference semantics (minddet/models/...):
//   NmsGpu / NmsNormalGpu     centerpoint/det3d_ms/ops/test_custom_pytorch/iou3d_nms_kernel.cu:491-601
//                             (kernels centerpoint/det3d_ms/ops/iou3d_nms/src/iou3d_nms_kernel.cu:267-372,
//                              host bitmask reduction iou3d_nms.cpp:102-133)
//   boxes_iou_nms_gpu         centerpoint/det3d_ms/ops/iou-bev-nms-org.cpp:237-283
//   BoxesIouBevGpu/OverlapBev iou3d_nms_kernel.cu:236-265
//   md_iou_aligned            pointpillars/src/core/box_np_ops.py:639-679
//   md_nms_aligned            pointpillars/src/core/nms.py:7-41,85-112
//   md_circle_nms             centerpoint/det3d_ms/core/utils/circle_nms_jit.py:6-36
//
// MI355X design (not a translation of the CUDA file):
//   * 64x64 suppression tiles are owned by a 256-thread workgroup = 4 wave64; lane = COLUMN
//     box (held in registers), the row box is an LDS broadcast, and the 64-bit mask word of a
//     (row, column-block) is ONE __ballot over the wave -- no per-lane bit loop, no shifts.
//     Only the upper-triangular tiles are launched.
//   * the greedy pass that the reference runs on the HOST after a blocking D2H copy of the
//     whole mask runs ON DEVICE in one workgroup: wave 0 resolves a 64-box diagonal block
//     with scalar bit tricks (one iteration per KEPT box, not per box), then the four waves
//     OR-reduce the kept rows' words into the "removed" words of later column blocks with
//     cross-lane DPP reductions.  No sync, no copies: the op is stream-ordered.
//   * batched (image, class)-keyed NMS: grid.y = list index, optional int32 group key.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"
#include "rot_geom.h"

#pragma clang fp contract(off)

namespace md {

constexpr int TILE = 64;

__global__ void rot_prep_kernel(const float *__restrict__ boxes, int n, float *__restrict__ rec) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float r[ROT_REC];
    float b[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) b[k] = boxes[(size_t)i * 7 + k];
    rot_make_record(b, r);
#pragma unroll
    for (int k = 0; k < ROT_REC; ++k) rec[(size_t)i * ROT_REC + k] = r[k];
}

// linear upper-triangular tile id -> (row block, col block)
__device__ __forceinline__ void tri_tile(int t, int cb, int &rb, int &cbk) {
    int r = 0;
    while (t >= cb - r) { t -= cb - r; ++r; }
    rb = r;
    cbk = r + t;
}

// MODE 0: IoU = so / fmaxf(sa+sb-so, 1e-8), suppress iff >  thr   (NmsGpu)
// MODE 1: ovr = so / (sa+sb-so),            suppress iff >= thr   (boxes_iou_nms_cpu)
template <int MODE>
__global__ __launch_bounds__(256) void nms_rot_mask_kernel(const float *__restrict__ rec, int n,
                                                            const float *__restrict__ thr_p,
                                                            unsigned long long *__restrict__ mask, int cb) {
    __shared__ float row_rec[TILE * ROT_REC];
    __shared__ float poly[3 * ROT_PTS * 256];
    int rb, cbk;
    tri_tile(blockIdx.x, cb, rb, cbk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < TILE * ROT_REC; e += 256) {
        const int g = rb * TILE + e / ROT_REC;
        row_rec[e] = g < n ? rec[(size_t)g * ROT_REC + e % ROT_REC] : 0.f;
    }
    const int gc = cbk * TILE + lane;
    float col[ROT_REC];
#pragma unroll
    for (int k = 0; k < ROT_REC; ++k) col[k] = gc < n ? rec[(size_t)gc * ROT_REC + k] : 0.f;
    const float thr = *thr_p;
    __syncthreads();
    float *scratch = poly + tid;
    for (int rr = wave * 16; rr < wave * 16 + 16; ++rr) {
        const int gr = rb * TILE + rr;
        if (gr >= n) break;  // wave-uniform
        bool pred = false;
        if (gc < n && (cbk != rb || lane > rr)) {
            const float *A = row_rec + rr * ROT_REC;
            const float so = rot_overlap(A, col, scratch, 256);
            if (MODE == 0) {
                const float v = so / fmaxf(A[14] + col[14] - so, ROT_EPS);
                pred = v > thr;
            } else {
                const float v = so / (A[14] + col[14] - so);
                pred = v >= thr;
            }
        }
        const unsigned long long w = __ballot(pred);
        if (lane == 0) mask[(size_t)gr * cb + cbk] = w;
    }
}

// ------------------------------------------------------------------ axis-aligned variants
// footprint of a 7-float box, iou_normal (iou3d_nms_kernel.cu:314-325)
__device__ __forceinline__ float iou_normal7(const float *a, const float *b) {
    const float left = fmaxf(a[0] - a[3] / 2, b[0] - b[3] / 2), right = fminf(a[0] + a[3] / 2, b[0] + b[3] / 2);
    const float top = fmaxf(a[1] - a[4] / 2, b[1] - b[4] / 2), bottom = fminf(a[1] + a[4] / 2, b[1] + b[4] / 2);
    const float w = fmaxf(right - left, 0.f), h = fmaxf(bottom - top, 0.f);
    const float inter = w * h;
    const float sa = a[3] * a[4], sb = b[3] * b[4];
    return inter / fmaxf(sa + sb - inter, ROT_EPS);
}

__global__ __launch_bounds__(256) void nms_normal7_mask_kernel(const float *__restrict__ boxes, int n,
                                                                const float *__restrict__ thr_p,
                                                                unsigned long long *__restrict__ mask, int cb) {
    __shared__ float row_box[TILE * 8];
    int rb, cbk;
    tri_tile(blockIdx.x, cb, rb, cbk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < TILE * 7; e += 256) {
        const int g = rb * TILE + e / 7;
        row_box[(e / 7) * 8 + e % 7] = g < n ? boxes[(size_t)g * 7 + e % 7] : 0.f;
    }
    const int gc = cbk * TILE + lane;
    float col[7];
#pragma unroll
    for (int k = 0; k < 7; ++k) col[k] = gc < n ? boxes[(size_t)gc * 7 + k] : 0.f;
    const float thr = *thr_p;
    __syncthreads();
    for (int rr = wave * 16; rr < wave * 16 + 16; ++rr) {
        const int gr = rb * TILE + rr;
        if (gr >= n) break;
        bool pred = false;
        if (gc < n && (cbk != rb || lane > rr)) pred = iou_normal7(row_box + rr * 8, col) > thr;
        const unsigned long long w = __ballot(pred);
        if (lane == 0) mask[(size_t)gr * cb + cbk] = w;
    }
}

// does row box a (kept, higher score) suppress column box c?
__device__ __forceinline__ bool aligned_suppresses(const float *a, const float4 c, float area_c, float off, int mode, float thr) {
    const float ax1 = a[0], ay1 = a[1], ax2 = a[2], ay2 = a[3];
    const float area_a = (ax2 - ax1 + off) * (ay2 - ay1 + off);
    const float w = rhi(rlo(ax2, c.z) - rhi(ax1, c.x) + off, 0.0f);
    const float h = rhi(rlo(ay2, c.w) - rhi(ay1, c.y) + off, 0.0f);
    const float inter = w * h;
    float ovr;
    if (mode == 2) ovr = inter / fmaxf(area_a + area_c - inter, ROT_EPS);
    else ovr = inter / (area_a + area_c - inter);
    return mode == 0 ? (ovr >= thr) : (ovr > thr);
}

// corner boxes [x1,y1,x2,y2]; modes documented in minddet_hip.h (md_nms_attrs)
__global__ __launch_bounds__(256) void nms_aligned_mask_kernel(const float *__restrict__ boxes_all,
                                                                const int *__restrict__ count,
                                                                const int *__restrict__ group_all, int n_max,
                                                                float thr, float eps, int mode,
                                                                unsigned long long *__restrict__ mask_all, int cb,
                                                                int cb_tri, int n_limit, const int *__restrict__ gate) {
    // cb = words per mask row (row stride); the launch covers the upper triangle of the first cb_tri x cb_tri tiles and the first
    // n_limit boxes of a list (the quota prefix pass of md_nms_aligned; cb_tri = cb, n_limit = n_max for a full pass); gate: the
    // full pass behind a prefix pass runs only for the lists whose flag is set
    __shared__ float row_box[TILE * 4];
    __shared__ int row_grp[TILE];
    const int list = blockIdx.y;
    if (gate && gate[list] == 0) return;
    const int n = min(count ? min(count[list], n_max) : n_max, n_limit);
    int rb, cbk;
    tri_tile(blockIdx.x, cb_tri, rb, cbk);
    if (rb * TILE >= n) return;  // block-uniform: nothing valid in this row block
    const float *boxes = boxes_all + (size_t)list * n_max * 4;
    const int *group = group_all ? group_all + (size_t)list * n_max : nullptr;
    unsigned long long *mask = mask_all + (size_t)list * n_max * cb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < TILE) {
        const int g = rb * TILE + tid;
        float4 v = g < n ? *reinterpret_cast<const float4 *>(boxes + (size_t)g * 4) : make_float4(0, 0, 0, 0);
        row_box[tid * 4 + 0] = v.x; row_box[tid * 4 + 1] = v.y; row_box[tid * 4 + 2] = v.z; row_box[tid * 4 + 3] = v.w;
        row_grp[tid] = (group && g < n) ? group[g] : 0;
    }
    const int gc = cbk * TILE + lane;
    const float4 c = gc < n ? *reinterpret_cast<const float4 *>(boxes + (size_t)gc * 4) : make_float4(0, 0, 0, 0);
    const int cg = (group && gc < n) ? group[gc] : 0;
    const float off = mode == 1 ? 1.0f : eps;
    const float area_c = (c.z - c.x + off) * (c.w - c.y + off);
    __syncthreads();
    for (int rr = wave * 16; rr < wave * 16 + 16; ++rr) {
        const int gr = rb * TILE + rr;
        if (gr >= n) break;
        bool pred = false;
        if (gc < n && (cbk != rb || lane > rr) && row_grp[rr] == cg)
            pred = aligned_suppresses(row_box + rr * 4, c, area_c, off, mode, thr);
        const unsigned long long w64 = __ballot(pred);
        if (lane == 0) mask[(size_t)gr * cb + cbk] = w64;
    }
}

__global__ __launch_bounds__(256) void circle_mask_kernel(const float *__restrict__ xy, int n,
                                                           const float *__restrict__ thr_p,
                                                           unsigned long long *__restrict__ mask, int cb) {
    __shared__ float row_xy[TILE * 2];
    int rb, cbk;
    tri_tile(blockIdx.x, cb, rb, cbk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < TILE * 2) {
        const int g = rb * TILE + tid / 2;
        row_xy[tid] = g < n ? xy[(size_t)g * 2 + (tid & 1)] : 0.f;
    }
    const int gc = cbk * TILE + lane;
    const float cx = gc < n ? xy[(size_t)gc * 2] : 0.f, cy = gc < n ? xy[(size_t)gc * 2 + 1] : 0.f;
    const float thr = *thr_p;
    __syncthreads();
    for (int rr = wave * 16; rr < wave * 16 + 16; ++rr) {
        const int gr = rb * TILE + rr;
        if (gr >= n) break;
        bool pred = false;
        if (gc < n && (cbk != rb || lane > rr)) {
            const float dx = row_xy[rr * 2] - cx, dy = row_xy[rr * 2 + 1] - cy;
            pred = (dx * dx + dy * dy) <= thr;
        }
        const unsigned long long w = __ballot(pred);
        if (lane == 0) mask[(size_t)gr * cb + cbk] = w;
    }
}

// ------------------------------------------------------------------ soft-NMS
// Soft-NMS (Bodla et al. 2017) as used by the reference through the un-vendored CenterNet Cython module
// (call site centernet/src/post_process.py:45-52: method 2 = gaussian, Nt 0.5, threshold 0.001, sigma 0.5, +1 pixel
// areas).  One wave per list (<= 1024 boxes, CenterNet has <= 100 per class): pick the best live box, decay every
// other live box by its overlap with it, drop those whose score falls below the threshold, repeat.
// method: 1 linear, 2 gaussian, 3 hard (weight 0 above Nt).
__global__ __launch_bounds__(64) void soft_nms_kernel(const float *__restrict__ boxes_all, const float *__restrict__ scores_all,
                                                       const int *__restrict__ count, int n_max, float sigma, float Nt,
                                                       float threshold, int method, float *__restrict__ out_scores_all,
                                                       int *__restrict__ order_all, int *__restrict__ num_all) {
    constexpr int PER = 16;  // boxes per lane -> up to 1024 per list
    const int list = blockIdx.x, lane = threadIdx.x;
    const int n = count ? min(count[list], n_max) : n_max;
    const float *boxes = boxes_all + (size_t)list * n_max * 4;
    float x1[PER], y1[PER], x2[PER], y2[PER], sc[PER];
    int state[PER];  // 0 live, 1 selected, 2 removed / absent
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int i = lane + 64 * t;
        state[t] = 2; sc[t] = 0.f; x1[t] = y1[t] = x2[t] = y2[t] = 0.f;
        if (i < n) {
            const float4 b = *reinterpret_cast<const float4 *>(boxes + (size_t)i * 4);
            x1[t] = b.x; y1[t] = b.y; x2[t] = b.z; y2[t] = b.w;
            sc[t] = scores_all[(size_t)list * n_max + i];
            state[t] = 0;
        }
    }
    int picked = 0;
    for (int it = 0; it < n; ++it) {
        // arg-max over live boxes (ties -> lowest index)
        float best = -3.0e38f;
        int bi = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < PER; ++t)
            if (state[t] == 0 && (sc[t] > best)) { best = sc[t]; bi = lane + 64 * t; }
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(best, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if (bi == 0x7fffffff) break;  // nothing live (wave-uniform)
        const int owner = bi & 63, slot = bi >> 6;
        float tx1 = 0, ty1 = 0, tx2 = 0, ty2 = 0;
#pragma unroll
        for (int t = 0; t < PER; ++t)
            if (t == slot) { tx1 = x1[t]; ty1 = y1[t]; tx2 = x2[t]; ty2 = y2[t]; if (lane == owner) state[t] = 1; }
        tx1 = __shfl(tx1, owner, 64); ty1 = __shfl(ty1, owner, 64); tx2 = __shfl(tx2, owner, 64); ty2 = __shfl(ty2, owner, 64);
        if (lane == 0) order_all[(size_t)list * n_max + picked] = bi;
        ++picked;
        const float tarea = (tx2 - tx1 + 1.f) * (ty2 - ty1 + 1.f);
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            if (state[t] != 0) continue;
            const float iw = fminf(tx2, x2[t]) - fmaxf(tx1, x1[t]) + 1.f;
            if (iw > 0.f) {
                const float ih = fminf(ty2, y2[t]) - fmaxf(ty1, y1[t]) + 1.f;
                if (ih > 0.f) {
                    const float area = (x2[t] - x1[t] + 1.f) * (y2[t] - y1[t] + 1.f);
                    const float ov = iw * ih / (tarea + area - iw * ih);
                    float w = 1.f;
                    if (method == 1) w = ov > Nt ? 1.f - ov : 1.f;
                    else if (method == 2) w = expf(-(ov * ov) / sigma);
                    else w = ov > Nt ? 0.f : 1.f;
                    sc[t] = w * sc[t];
                    if (sc[t] < threshold) state[t] = 2;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int i = lane + 64 * t;
        if (i < n_max) out_scores_all[(size_t)list * n_max + i] = state[t] == 1 ? sc[t] : 0.f;
    }
    for (int i = picked + lane; i < n_max; i += 64) order_all[(size_t)list * n_max + i] = 0;
    if (lane == 0) num_all[list] = picked;
}

// ------------------------------------------------------------------ on-device greedy pass
__device__ __forceinline__ unsigned long long uniform64(unsigned long long v) {
    const unsigned int lo = __builtin_amdgcn_readfirstlane((unsigned int)v);
    const unsigned int hi = __builtin_amdgcn_readfirstlane((unsigned int)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

constexpr int SCAN_KEEP_CAP = 4096;   // kept rows whose indices nms_scan_kernel holds in LDS (later ones are read back from its keep output)
__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v |= __shfl_xor(v, o, 64);
    return v;
}

// One workgroup per list.  mask rows are valid for column blocks >= the row's block.
// dead_area: optional per-box area array (stride in floats) -- boxes with area == 0 are
// removed up front without suppressing anything (iou-bev-nms-org.cpp:250-256).
template <typename KeepT>
__global__ __launch_bounds__(256) void nms_scan_kernel(const unsigned long long *__restrict__ mask_all,
                                                        const int *__restrict__ count, int n_max, int cb,
                                                        const float *__restrict__ dead_area, int dead_stride,
                                                        int max_output, KeepT *__restrict__ keep_all,
                                                        int *__restrict__ num_all,
                                                        unsigned char *__restrict__ keepmask_all, int n_limit = 0x7fffffff,
                                                        const int *__restrict__ gate = nullptr,
                                                        int *__restrict__ need_full = nullptr) {
    // n_limit / gate / need_full: the quota prefix pass of md_nms_aligned (see there).  A prefix pass looks at the first n_limit
    // boxes only and raises need_full[list] when they did not fill the quota although the list goes on.
    // The removed-bits of a column block are taken when the scan REACHES the block: the OR over the rows kept so far of their words for
    // that block -- one batch of independent loads per block (256 threads, the kept rows' indices in LDS), then the 64-step bit loop of
    // wave 0.  (r04: the r01 form OR-ed every kept row into ALL later column blocks as soon as it was kept -- up to 16 dependent round
    // trips per wave and block, most of them for blocks a quota'd scan never reaches: 8 us per block on the one-stage detectors' lists.)
    extern __shared__ unsigned long long remv[];  // [0..3] per-wave partial ORs, [4] kept word broadcast (cb + 1 >= 5 words are allocated)
    __shared__ int s_keep[SCAN_KEEP_CAP];
    const int list = blockIdx.x;
    if (gate && gate[list] == 0) return;
    const int n_all = count ? min(count[list], n_max) : n_max;
    const int n = min(n_all, n_limit);
    const unsigned long long *mask = mask_all + (size_t)list * n_max * cb;
    KeepT *keep = keep_all + (size_t)list * n_max;
    unsigned char *keepmask = keepmask_all ? keepmask_all + (size_t)list * n_max : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < n_max; i += 256) {
        keep[i] = 0;
        if (keepmask) keepmask[i] = 0;
    }
    __syncthreads();
    const int nb = (n + TILE - 1) / TILE;
    int total = 0;  // kept so far (uniform)
    for (int blk = 0; blk < nb; ++blk) {
        if (max_output > 0 && total >= max_output) break;  // quota reached (uniform): nothing later can be kept
        const int base = blk * TILE;
        const int r = base + lane;
        unsigned long long diag = 0ull;
        bool dead = r >= n;
        if (wave == 0 && r < n) {   // requested together with the kept rows' words below
            diag = mask[(size_t)r * cb + blk];
            if (dead_area && dead_area[(size_t)r * dead_stride] == 0.f) dead = true;
        }
        unsigned long long v = 0ull;
        for (int k = tid; k < total; k += 256) {
            const int row = k < SCAN_KEEP_CAP ? s_keep[k] : (int)keep[k];
            v |= mask[(size_t)row * cb + blk];
        }
        v = wave_or64(v);
        if (lane == 0) remv[wave] = v;
        __syncthreads();
        if (wave == 0) {
            unsigned long long cur = uniform64(remv[0] | remv[1] | remv[2] | remv[3]) | __ballot(dead);
            unsigned long long kept = 0ull;
            unsigned long long cand = ~cur;
            int quota = max_output > 0 ? max_output - total : 64;
            while (cand != 0ull && quota > 0) {
                const int i = __ffsll((long long)cand) - 1;
                kept |= 1ull << i;
                --quota;
                const unsigned int lo = __builtin_amdgcn_readlane((unsigned int)diag, i);
                const unsigned int hi = __builtin_amdgcn_readlane((unsigned int)(diag >> 32), i);
                cur |= ((unsigned long long)hi << 32) | lo | (1ull << i);
                cand = ~cur;
            }
            if ((kept >> lane) & 1ull) {
                const int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
                keep[pos] = (KeepT)r;
                if (pos < SCAN_KEEP_CAP) s_keep[pos] = r;
                if (keepmask) keepmask[r] = 1;
            }
            if (lane == 0) remv[4] = kept;
        }
        __syncthreads();   // (also makes this block's keep[] entries visible to the whole workgroup: one CU, one L1)
        total += __popcll(uniform64(remv[4]));
    }
    if (tid == 0) {
        num_all[list] = total;
        if (need_full) need_full[list] = (max_output > 0 && total < max_output && n_all > n) ? 1 : 0;
    }
}

// ------------------------------------------------------------------ IoU matrices
template <int IOU>
__global__ __launch_bounds__(256) void rot_pair_matrix_kernel(const float *__restrict__ rec_a, int na,
                                                               const float *__restrict__ rec_b, int nb,
                                                               float *__restrict__ out) {
    __shared__ float poly[3 * ROT_PTS * 256];
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t total = (size_t)na * nb;
    if (idx >= total) return;
    const int i = (int)(idx / nb), j = (int)(idx % nb);
    float A[ROT_REC], B[ROT_REC];
#pragma unroll
    for (int k = 0; k < ROT_REC; ++k) {
        A[k] = rec_a[(size_t)i * ROT_REC + k];
        B[k] = rec_b[(size_t)j * ROT_REC + k];
    }
    const float so = rot_overlap(A, B, poly + threadIdx.x, 256);
    out[idx] = IOU ? so / fmaxf(A[14] + B[14] - so, ROT_EPS) : so;
}

// rotate_iou_kernel_eval (pointpillars/eval_gpu/rotate_iou.py:167-302): the numba.cuda N x K rotated IoU with the
// point-in-quadrilateral + segment-intersection + pseudo-angle insertion sort formulation and the `criterion`
// switch.  One lane per pair; polygon scratch in LDS, slot-major.
__device__ __forceinline__ void rie_corners(const float *rb, float *c) {
    const float a_cos = cosf(rb[4]), a_sin = sinf(rb[4]);
    const float cx[4] = {-rb[2] / 2, -rb[2] / 2, rb[2] / 2, rb[2] / 2};
    const float cy[4] = {-rb[3] / 2, rb[3] / 2, rb[3] / 2, -rb[3] / 2};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        c[2 * i] = a_cos * cx[i] + a_sin * cy[i] + rb[0];
        c[2 * i + 1] = -a_sin * cx[i] + a_cos * cy[i] + rb[1];
    }
}
__device__ __forceinline__ bool rie_in_quad(float px, float py, const float *c) {
    const float ab0 = c[2] - c[0], ab1 = c[3] - c[1], ad0 = c[6] - c[0], ad1 = c[7] - c[1];
    const float ap0 = px - c[0], ap1 = py - c[1];
    const float abab = ab0 * ab0 + ab1 * ab1, abap = ab0 * ap0 + ab1 * ap1;
    const float adad = ad0 * ad0 + ad1 * ad1, adap = ad0 * ap0 + ad1 * ap1;
    return abab >= abap && abap >= 0 && adad >= adap && adap >= 0;
}
__global__ __launch_bounds__(256) void rotate_iou_eval_kernel(const float *__restrict__ boxes, int n,
                                                               const float *__restrict__ query, int k, int criterion,
                                                               float *__restrict__ out) {
    __shared__ float scratch[3 * 24 * 256];
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * k) return;
    const int i0 = (int)(idx / k), j0 = (int)(idx % k);
    float r1[5], r2[5], c1[8], c2[8];
#pragma unroll
    for (int t = 0; t < 5; ++t) { r1[t] = boxes[(size_t)i0 * 5 + t]; r2[t] = query[(size_t)j0 * 5 + t]; }
    rie_corners(r1, c1);
    rie_corners(r2, c2);
    float *px = scratch + threadIdx.x, *py = px + 24 * 256, *vs = py + 24 * 256;
    int m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (rie_in_quad(c1[2 * i], c1[2 * i + 1], c2)) { px[m * 256] = c1[2 * i]; py[m * 256] = c1[2 * i + 1]; ++m; }
        if (rie_in_quad(c2[2 * i], c2[2 * i + 1], c1)) { px[m * 256] = c2[2 * i]; py[m * 256] = c2[2 * i + 1]; ++m; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float A0 = c1[2 * i], A1 = c1[2 * i + 1], B0 = c1[2 * ((i + 1) & 3)], B1 = c1[2 * ((i + 1) & 3) + 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float C0 = c2[2 * j], C1 = c2[2 * j + 1], D0 = c2[2 * ((j + 1) & 3)], D1 = c2[2 * ((j + 1) & 3) + 1];
            const float BA0 = B0 - A0, BA1 = B1 - A1, DA0 = D0 - A0, CA0 = C0 - A0, DA1 = D1 - A1, CA1 = C1 - A1;
            const bool acd = DA1 * CA0 > CA1 * DA0;
            const bool bcd = (D1 - B1) * (C0 - B0) > (C1 - B1) * (D0 - B0);
            if (acd != bcd) {
                const bool abc = CA1 * BA0 > BA1 * CA0, abd = DA1 * BA0 > BA1 * DA0;
                if (abc != abd && m < 24) {
                    const float DC0 = D0 - C0, DC1 = D1 - C1;
                    const float ABBA = A0 * B1 - B0 * A1, CDDC = C0 * D1 - D0 * C1;
                    const float DH = BA1 * DC0 - BA0 * DC1;
                    px[m * 256] = (ABBA * DC0 - BA0 * CDDC) / DH;
                    py[m * 256] = (ABBA * DC1 - BA1 * CDDC) / DH;
                    ++m;
                }
            }
        }
    }
    float ai = 0.f;
    if (m > 0) {
        float cx = 0.f, cy = 0.f;
        for (int i = 0; i < m; ++i) { cx += px[i * 256]; cy += py[i * 256]; }
        cx /= (float)m;
        cy /= (float)m;
        for (int i = 0; i < m; ++i) {
            float v0 = px[i * 256] - cx, v1 = py[i * 256] - cy;
            const float d = sqrtf(v0 * v0 + v1 * v1);
            v0 = v0 / d;
            v1 = v1 / d;
            if (v1 < 0) v0 = -2 - v0;
            vs[i * 256] = v0;
        }
        for (int i = 1; i < m; ++i) {
            if (vs[(i - 1) * 256] > vs[i * 256]) {
                const float temp = vs[i * 256], tx = px[i * 256], ty = py[i * 256];
                int j = i;
                while (j > 0 && vs[(j - 1) * 256] > temp) {
                    vs[j * 256] = vs[(j - 1) * 256];
                    px[j * 256] = px[(j - 1) * 256];
                    py[j * 256] = py[(j - 1) * 256];
                    --j;
                }
                vs[j * 256] = temp; px[j * 256] = tx; py[j * 256] = ty;
            }
        }
        const float ax = px[0], ay = py[0];
        for (int i = 0; i < m - 2; ++i) {
            const float bx = px[(i + 1) * 256], by = py[(i + 1) * 256], qx = px[(i + 2) * 256], qy = py[(i + 2) * 256];
            ai += fabsf(((ax - qx) * (by - qy) - (ay - qy) * (bx - qx)) / 2.0f);
        }
    }
    const float a1 = r1[2] * r1[3], a2 = r2[2] * r2[3];
    float v;
    if (criterion == -1) v = ai / (a1 + a2 - ai);
    else if (criterion == 0) v = ai / a1;
    else if (criterion == 1) v = ai / a2;
    else v = ai;
    out[idx] = v;
}

__global__ void iou_aligned_kernel(const float *__restrict__ boxes, int n, const float *__restrict__ query, int k,
                                   float eps, float *__restrict__ out) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * k) return;
    const int i = (int)(idx / k), q = (int)(idx % k);
    const float4 b = *reinterpret_cast<const float4 *>(boxes + (size_t)i * 4);
    const float4 c = *reinterpret_cast<const float4 *>(query + (size_t)q * 4);
    const float qa = (c.z - c.x + eps) * (c.w - c.y + eps);
    float v = 0.f;
    const float iw = rlo(b.z, c.z) - rhi(b.x, c.x) + eps;
    if (iw > 0) {
        const float ih = rlo(b.w, c.w) - rhi(b.y, c.y) + eps;
        if (ih > 0) {
            const float ua = (b.z - b.x + eps) * (b.w - b.y + eps) + qa - iw * ih;
            v = iw * ih / ua;
        }
    }
    out[idx] = v;
}

static inline size_t scan_lds(int cb) { return (size_t)(cb + 1 < 5 ? 5 : cb + 1) * sizeof(unsigned long long); }

}  // namespace md

using namespace md;

extern "C" const char *md_version(void) { return "minddet_hip 0.1 gfx950"; }
extern "C" int md_scratch_release(void) { return md::pool_release(); }

static int check_boxes7(int nparam, int want, void **params, int *ndims, int64_t **shapes, const char **dtypes,
                        int64_t &n) {
    if (nparam != want && nparam != want + 1) return MD_ERR_NPARAM;
    if (!params || !dtype_is(dtypes, 0, "float32")) return MD_ERR_ARG;
    n = dim(ndims, shapes, 0, 0);
    if (n < 0 || dim(ndims, shapes, 0, 1) != 7) return MD_ERR_ARG;
    if (n > (1 << 16)) return MD_ERR_SIZE;
    return MD_OK;
}

// outputs of the keep-list NMS ops: keep[>= n] and num[>= 1] (the kernels zero keep[0..n) and write num[0] whatever the
// caller's out_shape lambdas said), no NULL operand when there is work.  Shapes a caller does not describe are not checked.
static int check_keep_outputs(void **params, int *ndims, int64_t **shapes, int64_t n) {
    const int64_t keep_n = numel(ndims, shapes, 2), num_n = numel(ndims, shapes, 3);
    if ((keep_n >= 0 && keep_n < n) || (num_n >= 0 && num_n < 1)) return MD_ERR_ARG;
    if (!params[3]) return MD_ERR_ARG;
    if (n > 0 && (!params[0] || !params[1] || !params[2])) return MD_ERR_ARG;
    return MD_OK;
}

template <int MODE, typename KeepT>
static int rot_nms_impl(MD_AOT_ARGS, const char *keep_dtype) {
    int64_t n;
    int rc = check_boxes7(nparam, 4, params, ndims, shapes, dtypes, n);
    if (rc) return rc;
    if (!dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, keep_dtype) || !dtype_is(dtypes, 3, "int32"))
        return MD_ERR_ARG;
    rc = check_keep_outputs(params, ndims, shapes, n);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    KeepT *keep = (KeepT *)params[2];
    int *num = (int *)params[3];
    if (n == 0) return hipMemsetAsync(num, 0, sizeof(int), s) == hipSuccess ? MD_OK : MD_ERR_HIP;
    const int cb = (int)((n + TILE - 1) / TILE);
    const size_t rec_bytes = align_up((size_t)n * ROT_REC * 4, 256);
    const size_t mask_bytes = (size_t)n * cb * 8;
    Scratch ws;
    rc = ws.acquire(rec_bytes + mask_bytes, nparam, params, ndims, shapes, 4, s);
    if (rc) return rc;
    float *rec = (float *)ws.ptr;
    unsigned long long *mask = (unsigned long long *)((char *)ws.ptr + rec_bytes);
    hipLaunchKernelGGL(rot_prep_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const float *)params[0], (int)n, rec);
    hipLaunchKernelGGL((nms_rot_mask_kernel<MODE>), dim3(cb * (cb + 1) / 2), dim3(256), 0, s, rec, (int)n,
                       (const float *)params[1], mask, cb);
    hipLaunchKernelGGL((nms_scan_kernel<KeepT>), dim3(1), dim3(256), scan_lds(cb), s, mask, (const int *)nullptr,
                       (int)n, cb, MODE == 1 ? rec + 14 : (const float *)nullptr, ROT_REC, 0, keep, num,
                       (unsigned char *)nullptr);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int NmsGpu(MD_AOT_ARGS) {
    return rot_nms_impl<0, long long>(nparam, params, ndims, shapes, dtypes, stream, extra, "int64");
}

extern "C" int boxes_iou_nms_gpu(MD_AOT_ARGS) {
    return rot_nms_impl<1, int>(nparam, params, ndims, shapes, dtypes, stream, extra, "int32");
}

extern "C" int NmsNormalGpu(MD_AOT_ARGS) {
    int64_t n;
    int rc = check_boxes7(nparam, 4, params, ndims, shapes, dtypes, n);
    if (rc) return rc;
    if (!dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int64") || !dtype_is(dtypes, 3, "int32"))
        return MD_ERR_ARG;
    rc = check_keep_outputs(params, ndims, shapes, n);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return hipMemsetAsync(params[3], 0, sizeof(int), s) == hipSuccess ? MD_OK : MD_ERR_HIP;
    const int cb = (int)((n + TILE - 1) / TILE);
    Scratch ws;
    rc = ws.acquire((size_t)n * cb * 8, nparam, params, ndims, shapes, 4, s);
    if (rc) return rc;
    unsigned long long *mask = (unsigned long long *)ws.ptr;
    hipLaunchKernelGGL(nms_normal7_mask_kernel, dim3(cb * (cb + 1) / 2), dim3(256), 0, s, (const float *)params[0],
                       (int)n, (const float *)params[1], mask, cb);
    hipLaunchKernelGGL((nms_scan_kernel<long long>), dim3(1), dim3(256), scan_lds(cb), s, mask, (const int *)nullptr,
                       (int)n, cb, (const float *)nullptr, 0, 0, (long long *)params[2], (int *)params[3],
                       (unsigned char *)nullptr);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

template <int IOU>
static int rot_matrix_impl(MD_AOT_ARGS) {
    if (nparam != 3 && nparam != 4) return MD_ERR_NPARAM;
    if (!params || !dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "float32"))
        return MD_ERR_ARG;
    const int64_t na = dim(ndims, shapes, 0, 0), nb = dim(ndims, shapes, 1, 0);
    if (na < 0 || nb < 0 || dim(ndims, shapes, 0, 1) != 7 || dim(ndims, shapes, 1, 1) != 7) return MD_ERR_ARG;
    if (na == 0 || nb == 0) return MD_OK;
    if (na > (1 << 24) || nb > (1 << 24)) return MD_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    const size_t ra = align_up((size_t)na * ROT_REC * 4, 256), rbb = (size_t)nb * ROT_REC * 4;
    Scratch ws;
    int rc = ws.acquire(ra + rbb, nparam, params, ndims, shapes, 3, s);
    if (rc) return rc;
    float *rec_a = (float *)ws.ptr, *rec_b = (float *)((char *)ws.ptr + ra);
    hipLaunchKernelGGL(rot_prep_kernel, dim3((na + 255) / 256), dim3(256), 0, s, (const float *)params[0], (int)na, rec_a);
    hipLaunchKernelGGL(rot_prep_kernel, dim3((nb + 255) / 256), dim3(256), 0, s, (const float *)params[1], (int)nb, rec_b);
    const size_t total = (size_t)na * nb;
    hipLaunchKernelGGL((rot_pair_matrix_kernel<IOU>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, rec_a,
                       (int)na, rec_b, (int)nb, (float *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int BoxesIouBevGpu(MD_AOT_ARGS) {
    return rot_matrix_impl<1>(nparam, params, ndims, shapes, dtypes, stream, extra);
}
extern "C" int BoxesOverlapBevGpu(MD_AOT_ARGS) {
    return rot_matrix_impl<0>(nparam, params, ndims, shapes, dtypes, stream, extra);
}

extern "C" int md_iou_aligned(MD_AOT_ARGS) {
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "float32"))
        return MD_ERR_ARG;
    const int64_t n = dim(ndims, shapes, 0, 0), k = dim(ndims, shapes, 1, 0);
    if (n < 0 || k < 0 || dim(ndims, shapes, 0, 1) != 4 || dim(ndims, shapes, 1, 1) != 4) return MD_ERR_ARG;
    if (n == 0 || k == 0) return MD_OK;
    const float eps = extra ? ((const md_iou_attrs *)extra)->eps : 0.f;
    const size_t total = (size_t)n * k;
    hipLaunchKernelGGL(iou_aligned_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (int)n, (const float *)params[1], (int)k, eps, (float *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_rotate_iou_eval(MD_AOT_ARGS) {
    // in: boxes[N,5] f32, query[K,5] f32 ; out: iou[N,K] f32.  extra: md_rotate_iou_attrs (NULL -> criterion -1)
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "float32"))
        return MD_ERR_ARG;
    const int64_t n = dim(ndims, shapes, 0, 0), k = dim(ndims, shapes, 1, 0);
    if (n < 0 || k < 0 || dim(ndims, shapes, 0, 1) != 5 || dim(ndims, shapes, 1, 1) != 5) return MD_ERR_ARG;
    if (n == 0 || k == 0) return MD_OK;
    if (n > (1 << 24) || k > (1 << 24)) return MD_ERR_SIZE;
    const int criterion = extra ? ((const md_rotate_iou_attrs *)extra)->criterion : -1;
    const size_t total = (size_t)n * k;
    hipLaunchKernelGGL(rotate_iou_eval_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (int)n, (const float *)params[1], (int)k, criterion, (float *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_nms_aligned(MD_AOT_ARGS) {
    if (nparam != 6 && nparam != 7) return MD_ERR_NPARAM;
    if (!params || !extra) return MD_ERR_ARG;  // params[0] may be a null pointer for an empty tensor
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "int32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "uint8") || !dtype_is(dtypes, 4, "int32") || !dtype_is(dtypes, 5, "int32"))
        return MD_ERR_ARG;
    const int nd = ndims ? ndims[0] : -1;
    if (nd != 2 && nd != 3) return MD_ERR_ARG;
    const int64_t B = nd == 3 ? shapes[0][0] : 1, n = shapes[0][nd - 2];
    if (shapes[0][nd - 1] != 4 || B < 0 || n < 0) return MD_ERR_ARG;
    if (n > (1 << 16) || B > 65535) return MD_ERR_SIZE;
    const md_nms_attrs *at = (const md_nms_attrs *)extra;
    if (at->mode < 0 || at->mode > 2) return MD_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (B == 0) return MD_OK;
    if (n == 0) return hipMemsetAsync(params[5], 0, sizeof(int) * B, s) == hipSuccess ? MD_OK : MD_ERR_HIP;
    if (!params[0]) return MD_ERR_ARG;
    const int cb = (int)((n + TILE - 1) / TILE);
    const size_t mask_bytes = (size_t)B * n * cb * 8;
    // Quota prefix pass.  With an output quota (max_output > 0) the scan stops at the quota-th kept box, and a box is suppressed by
    // higher-ranked kept boxes only: the first P boxes decide among themselves, so when they already yield max_output survivors the
    // P x P corner of the mask is all that was ever needed (YOLO: 4 096 candidates, quota 300 -> 190 tiles instead of 2 080 per
    // image).  Pass 1 = mask corner + scan over the first P boxes; it raises a per-list flag when the quota was NOT filled although
    // the list goes on, and pass 2 (full mask + full scan, gated by that flag on the device: no host synchronisation) redoes those
    // lists from scratch.  Outputs are identical to the single full pass in every case.
    const int64_t want = (int64_t)4 * at->max_output > 512 ? (int64_t)4 * at->max_output : 512;
    const int P = (int)((want + TILE - 1) / TILE * TILE);
    const bool two_level = at->max_output > 0 && n >= 2 * (int64_t)P;
    const size_t flag_bytes = two_level ? align_up((size_t)B * 4, 256) : 0;
    Scratch ws;
    int rc = ws.acquire(mask_bytes + flag_bytes, nparam, params, ndims, shapes, 6, s);
    bool tl = two_level;
    if (rc == MD_ERR_SIZE && two_level) {   // a caller workspace sized for the mask only: single full pass
        tl = false;
        rc = ws.acquire(mask_bytes, nparam, params, ndims, shapes, 6, s);
    }
    if (rc) return rc;
    unsigned long long *mask = (unsigned long long *)ws.ptr;
    int *need_full = nullptr;
    if (tl) {
        need_full = (int *)((char *)ws.ptr + mask_bytes);   // (mask_bytes is a multiple of 8)
        const int cbp = P / TILE;
        hipLaunchKernelGGL(nms_aligned_mask_kernel, dim3(cbp * (cbp + 1) / 2, (unsigned)B), dim3(256), 0, s,
                           (const float *)params[0], (const int *)params[1], (const int *)params[2], (int)n,
                           at->iou_threshold, at->eps, at->mode, mask, cb, cbp, P, (const int *)nullptr);
        hipLaunchKernelGGL((nms_scan_kernel<int>), dim3((unsigned)B), dim3(256), scan_lds(cb), s, mask,
                           (const int *)params[1], (int)n, cb, (const float *)nullptr, 0, at->max_output, (int *)params[4],
                           (int *)params[5], (unsigned char *)params[3], P, (const int *)nullptr, need_full);
    }
    hipLaunchKernelGGL(nms_aligned_mask_kernel, dim3(cb * (cb + 1) / 2, (unsigned)B), dim3(256), 0, s,
                       (const float *)params[0], (const int *)params[1], (const int *)params[2], (int)n,
                       at->iou_threshold, at->eps, at->mode, mask, cb, cb, (int)n, (const int *)need_full);
    hipLaunchKernelGGL((nms_scan_kernel<int>), dim3((unsigned)B), dim3(256), scan_lds(cb), s, mask,
                       (const int *)params[1], (int)n, cb, (const float *)nullptr, 0, at->max_output, (int *)params[4],
                       (int *)params[5], (unsigned char *)params[3], 0x7fffffff, (const int *)need_full, (int *)nullptr);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_soft_nms(MD_AOT_ARGS) {
    // in: boxes[L,N,4] f32 (or [N,4]), scores[L,N] f32, count[L] i32 | NULL ; out: scores_out[L,N] f32 (0 = removed),
    //     order[L,N] i32 (selection order, leading num valid), num[L] i32.  extra: md_soft_nms_attrs
    if (nparam != 6) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "float32") || !dtype_is(dtypes, 4, "int32") || !dtype_is(dtypes, 5, "int32"))
        return MD_ERR_ARG;
    const int nd = ndims[0];
    if (nd != 2 && nd != 3) return MD_ERR_ARG;
    const int64_t L = nd == 3 ? shapes[0][0] : 1, n = shapes[0][nd - 2];
    if (shapes[0][nd - 1] != 4 || numel(ndims, shapes, 1) != L * n || numel(ndims, shapes, 3) != L * n ||
        numel(ndims, shapes, 4) != L * n || numel(ndims, shapes, 5) != L)
        return MD_ERR_ARG;
    if (n > 1024) return MD_ERR_SIZE;
    const md_soft_nms_attrs *at = (const md_soft_nms_attrs *)extra;
    if (at->method < 1 || at->method > 3 || !(at->sigma > 0.f)) return MD_ERR_ARG;
    if (L == 0) return MD_OK;
    hipLaunchKernelGGL(soft_nms_kernel, dim3((unsigned)L), dim3(64), 0, (hipStream_t)stream, (const float *)params[0],
                       (const float *)params[1], (const int *)params[2], (int)n, at->sigma, at->Nt, at->threshold, at->method,
                       (float *)params[3], (int *)params[4], (int *)params[5]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_circle_nms(MD_AOT_ARGS) {
    if (nparam != 5 && nparam != 6) return MD_ERR_NPARAM;
    if (!params || !dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "uint8") ||
        !dtype_is(dtypes, 3, "int32") || !dtype_is(dtypes, 4, "int32"))
        return MD_ERR_ARG;
    const int64_t n = dim(ndims, shapes, 0, 0);
    if (n < 0 || dim(ndims, shapes, 0, 1) != 2) return MD_ERR_ARG;
    if (n > (1 << 16)) return MD_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return hipMemsetAsync(params[4], 0, sizeof(int), s) == hipSuccess ? MD_OK : MD_ERR_HIP;
    const int cb = (int)((n + TILE - 1) / TILE);
    Scratch ws;
    int rc = ws.acquire((size_t)n * cb * 8, nparam, params, ndims, shapes, 5, s);
    if (rc) return rc;
    unsigned long long *mask = (unsigned long long *)ws.ptr;
    hipLaunchKernelGGL(circle_mask_kernel, dim3(cb * (cb + 1) / 2), dim3(256), 0, s, (const float *)params[0], (int)n,
                       (const float *)params[1], mask, cb);
    hipLaunchKernelGGL((nms_scan_kernel<int>), dim3(1), dim3(256), scan_lds(cb), s, mask, (const int *)nullptr, (int)n,
                       cb, (const float *)nullptr, 0, 0, (int *)params[3], (int *)params[4], (unsigned char *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}
