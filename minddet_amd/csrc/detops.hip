// detops.hip -- anchor/prior generation, box codecs, segmented top-k, RoIAlign, CenterNet decode.
//
// Reference counterparts (minddet/models/...):
//   md_anchors_3d_stride   pointpillars/src/core/box_np_ops.py:453-523 (create_anchors_3d_stride)
//   md_anchors_3d_range    pointpillars/src/core/box_np_ops.py:526-568 (create_anchors_3d_range); both write into the
//                          concatenated table of several generators (pointpillars/src/core/target_assigner.py:227-249)
//   md_anchor_mask         pointpillars/src/data/preprocess.py:211-225 +
//                          pointpillars/src/core/box_np_ops.py:745-776
//   md_second_box_decode   pointpillars/src/core/box_ops.py:47-85 / box_np_ops.py:40-67
//   md_topk_segmented      ops.TopK(sorted=True) call sites: centernet/src/decode.py:81,96,101;
//                          centerpoint/.../center_head.py:435; pointpillars/src/pointpillars.py:764
//                          (stated as a stable descending sort)
//   md_centernet_decode    centernet/src/decode.py:40-64,90-109,151-196 + utils.py:48-129
//   md_anchors_fpn, md_delta2bbox, md_roi_align: absent from the reference:
//                          public definitions (mmdet / torchvision), "parity unpinned".
//
// All of these are HBM- or latency-bound integer/float streaming work: coalesced 16-byte
// accesses, LDS histograms / bitonic sort for the select, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

#include "aot.h"

#pragma clang fp contract(off)

namespace md {

static inline unsigned grid1d(size_t total, int bs = 256) {
    size_t b = (total + bs - 1) / bs;
    return (unsigned)(b > 16384 ? 16384 : (b == 0 ? 1 : b));
}

// ------------------------------------------------------------------------------------------ anchors
// 2-D FPN anchors: out[(loc * A + a), 4] = shift(x,y) + base[level][a]; locations row-major.
struct FpnLevel { int H, W, stride; int64_t offset; };  // offset in anchors
struct FpnArgs { int L, A; FpnLevel lv[8]; float base[8][16][4]; };

__global__ void anchors_fpn_kernel(FpnArgs a, float4 *__restrict__ out, size_t total) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < a.L && (int64_t)e >= a.lv[l + 1].offset) ++l;
        const size_t r = e - a.lv[l].offset;
        const int an = (int)(r % a.A);
        const size_t loc = r / a.A;
        const int x = (int)(loc % a.lv[l].W), y = (int)(loc / a.lv[l].W);
        const float sx = (float)x * (float)a.lv[l].stride, sy = (float)y * (float)a.lv[l].stride;
        const float *b = a.base[l][an];
        out[e] = make_float4(sx + b[0], sy + b[1], sx + b[2], sy + b[3]);
    }
}

// PointPillars anchors, layout [1,H,W,S(=1 size slot per call),R,7]; np.arange float32 fill
// semantics: v[i] = first + float(i) * delta with delta = f32(second) - f32(first).
struct Anchor3dArgs {
    int slot_off, slots;   // rows [slot_off, slot_off + R) of the [.., slots, 7] anchor table of a location (concat of generators)
    int H, W, R;
    float x_first, x_delta, y_first, y_delta, z;
    float size[3];
    float rot[8];
};
__global__ void anchors_3d_stride_kernel(Anchor3dArgs a, float *__restrict__ out) {
    const size_t total = (size_t)a.H * a.W * a.R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % a.R);
        const size_t loc = e / a.R;
        const int x = (int)(loc % a.W), y = (int)(loc / a.W);
        float *o = out + (loc * a.slots + a.slot_off + r) * 7;
        o[0] = a.x_first + (float)x * a.x_delta;
        o[1] = a.y_first + (float)y * a.y_delta;
        o[2] = a.z;
        o[3] = a.size[0]; o[4] = a.size[1]; o[5] = a.size[2];
        o[6] = a.rot[r];
    }
}

// create_anchors_3d_range (pointpillars/src/core/box_np_ops.py:526-568): centres = np.linspace(lo, hi, n, dtype=float32) per axis.
// numpy's linspace: step = f32(f32(hi - lo) / (n - 1)), v[i] = i * step + lo, v[n-1] = hi exactly; the product and sum run
//   mode 0: in float32 (numpy >= 2: NEP 50 keeps float32 scalars float32 -- what the reference code computes on this image),
//   mode 1: in float64, rounded once to float32 (numpy 1.21, the reference's pin: result_type(f32, f32, float(num)) = float64).
struct LinAxis { float lo, hi, step; int n; };
struct Anchor3dRangeArgs {
    int D, H, W, S, R, mode, slot_off, slots;
    LinAxis ax[3];   // x, y, z
    float size[4][3];
    float rot[8];
};
__device__ __forceinline__ float linspace_at(const LinAxis &ax, int i, int mode) {
    if (ax.n == 1) return ax.lo;
    if (i == ax.n - 1) return ax.hi;
    if (ax.step == 0.f) {   // numpy's any_step_zero branch: (i / div) * delta + lo
        const float delta = ax.hi - ax.lo;
        if (mode == 0) return ((float)i / (float)(ax.n - 1)) * delta + ax.lo;
        return (float)(((double)i / (double)(ax.n - 1)) * (double)delta + (double)ax.lo);
    }
    if (mode == 0) return (float)i * ax.step + ax.lo;
    return (float)((double)i * (double)ax.step + (double)ax.lo);
}
__global__ void anchors_3d_range_kernel(Anchor3dRangeArgs a, float *__restrict__ out) {
    const size_t total = (size_t)a.D * a.H * a.W * a.S * a.R;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(e % a.R);
        size_t t = e / a.R;
        const int sz = (int)(t % a.S);
        const size_t loc = t / a.S;
        const int x = (int)(loc % a.W), y = (int)((loc / a.W) % a.H), z = (int)(loc / ((size_t)a.W * a.H));
        float *o = out + (loc * a.slots + a.slot_off + sz * a.R + r) * 7;
        o[0] = linspace_at(a.ax[0], x, a.mode);
        o[1] = linspace_at(a.ax[1], y, a.mode);
        o[2] = linspace_at(a.ax[2], z, a.mode);
        o[3] = a.size[sz][0]; o[4] = a.size[sz][1]; o[5] = a.size[sz][2];
        o[6] = a.rot[r];
    }
}

// anchor mask: dense count map -> integral image -> 4-corner box sums (integer exact)
__global__ void amask_scatter_kernel(const int *__restrict__ coors, int nv, int nx, int ny, int *__restrict__ dense) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    const int y = coors[i * 3 + 1], x = coors[i * 3 + 2];
    if ((unsigned)y < (unsigned)ny && (unsigned)x < (unsigned)nx) atomicAdd(&dense[y * nx + x], 1);
}
// cumsum along axis 0 (y) then axis 1 (x): one thread per column / per row (maps are ~500x500)
__global__ void amask_cumsum_y_kernel(int *__restrict__ dense, int nx, int ny) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= nx) return;
    int acc = 0;
    for (int y = 0; y < ny; ++y) { acc += dense[y * nx + x]; dense[y * nx + x] = acc; }
}
__global__ void amask_cumsum_x_kernel(int *__restrict__ dense, int nx, int ny) {
    // one wave per row, wave-level inclusive scan over 64-wide chunks
    const int y = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (y >= ny) return;
    int carry = 0;
    for (int x0 = 0; x0 < nx; x0 += 64) {
        const int x = x0 + lane;
        int v = x < nx ? dense[y * nx + x] : 0;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if (lane >= o) v += t;
        }
        v += carry;
        if (x < nx) dense[y * nx + x] = v;
        carry = __shfl(v, 63, 64);
    }
}
__global__ void amask_area_kernel(const int *__restrict__ dense, const float *__restrict__ bv, int n, int nx, int ny,
                                  float sx, float sy, float ox, float oy, float thr, float *__restrict__ area,
                                  unsigned char *__restrict__ mask) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 b = *reinterpret_cast<const float4 *>(bv + (size_t)i * 4);
    int c0 = (int)floorf((b.x - ox) / sx), c1 = (int)floorf((b.y - oy) / sy);
    int c2 = (int)floorf((b.z - ox) / sx), c3 = (int)floorf((b.w - oy) / sy);
    c0 = max(c0, 0); c1 = max(c1, 0); c2 = min(c2, nx - 1); c3 = min(c3, ny - 1);
    // numpy negative indices wrap (a box entirely left of / below the grid): mirror that
    const int C0 = c0, C1 = c1, C2 = c2 < 0 ? c2 + nx : c2, C3 = c3 < 0 ? c3 + ny : c3;
    const int c0w = min(C0, nx - 1), c1w = min(C1, ny - 1);
    const int v = dense[C3 * nx + C2] - dense[C3 * nx + c0w] - dense[c1w * nx + C2] + dense[c1w * nx + c0w];
    const float a = (float)v;
    area[i] = a;
    mask[i] = a > thr ? 1 : 0;
}

// ------------------------------------------------------------------------------------------ codecs
__global__ void second_box_decode_kernel(const float *__restrict__ enc, const float *__restrict__ anc, size_t n,
                                         size_t n_anchor, float *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float *t = enc + i * 7, *a = anc + (i % n_anchor) * 7;
        const float xa = a[0], ya = a[1], wa = a[3], la = a[4], ha = a[5], ra = a[6];
        const float za = a[2] + ha / 2;
        const float diagonal = sqrtf(la * la + wa * wa);
        const float xg = t[0] * diagonal + xa, yg = t[1] * diagonal + ya, zg = t[2] * ha + za;
        const float lg = expf(t[4]) * la, wg = expf(t[3]) * wa, hg = expf(t[5]) * ha;
        const float rg = t[6] + ra;
        float *o = out + i * 7;
        o[0] = xg; o[1] = yg; o[2] = zg - hg / 2; o[3] = wg; o[4] = lg; o[5] = hg; o[6] = rg;
    }
}

struct DeltaArgs { float mean[4], stdv[4]; float max_ratio; float clip_w, clip_h; int do_clip; };
// rois[n,4], deltas[n,4] (optionally gathered: idx[n] selects rows of rois_src/deltas_src)
__global__ void delta2bbox_kernel(const float *__restrict__ rois, const float *__restrict__ deltas, size_t n,
                                  DeltaArgs a, float *__restrict__ out) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 r = *reinterpret_cast<const float4 *>(rois + i * 4);
        const float4 d0 = *reinterpret_cast<const float4 *>(deltas + i * 4);
        const float dx = d0.x * a.stdv[0] + a.mean[0], dy = d0.y * a.stdv[1] + a.mean[1];
        float dw = d0.z * a.stdv[2] + a.mean[2], dh = d0.w * a.stdv[3] + a.mean[3];
        dw = fminf(fmaxf(dw, -a.max_ratio), a.max_ratio);
        dh = fminf(fmaxf(dh, -a.max_ratio), a.max_ratio);
        const float px = (r.x + r.z) * 0.5f, py = (r.y + r.w) * 0.5f, pw = r.z - r.x, ph = r.w - r.y;
        const float gw = pw * expf(dw), gh = ph * expf(dh);
        const float gx = px + pw * dx, gy = py + ph * dy;
        float x1 = gx - gw * 0.5f, y1 = gy - gh * 0.5f, x2 = gx + gw * 0.5f, y2 = gy + gh * 0.5f;
        if (a.do_clip) {
            x1 = fminf(fmaxf(x1, 0.f), a.clip_w); x2 = fminf(fmaxf(x2, 0.f), a.clip_w);
            y1 = fminf(fmaxf(y1, 0.f), a.clip_h); y2 = fminf(fmaxf(y2, 0.f), a.clip_h);
        }
        *reinterpret_cast<float4 *>(out + i * 4) = make_float4(x1, y1, x2, y2);
    }
}

// ------------------------------------------------------------------------------------------ segmented top-k
// order-preserving map float -> uint (ascending)
__device__ __forceinline__ unsigned ford(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float funord(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

#ifdef MD_DIAG
// tools/topk_stamps.py: cycle stamps of segment 0's workgroup (thread 0)
__device__ unsigned long long *g_topk_stamps = nullptr;
#define TK_STAMP(I) do { if (g_topk_stamps && blockIdx.x == 0 && threadIdx.x == 0) g_topk_stamps[I] = __builtin_readcyclecounter(); } while (0)
#else
#define TK_STAMP(I) do { } while (0)
#endif
constexpr int TOPK_THREADS = 1024;
constexpr int TOPK_MAXK = 4096;
constexpr int TOPK_LDS_MAX_N = 30000;   // longest segment whose keys topk_segmented_lds_kernel stages in LDS (and registers)

// Histogram increment with wave-level aggregation.  Detector scores are heavily concentrated (bf16
// logits: a handful of distinct top bytes), so a plain LDS atomic per lane serialises 64-deep on one
// address.  Lanes that share the leader's bucket are counted with one ballot and added once; after a
// few rounds the stragglers (spread-out data) fall back to per-lane atomics.
__device__ __forceinline__ void hist_add_aggregated(unsigned *hist, bool active, unsigned bucket) {
    const unsigned long long act = __ballot(active);
    if (act == 0ull) return;
    const int leader = __ffsll((long long)act) - 1;
    const unsigned b = __builtin_amdgcn_readlane(bucket, leader);
    const unsigned long long same = __ballot(active && bucket == b);
    if (same == act) {  // every active lane hits one bucket (degenerate digit): one add for the wave
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[b], (unsigned)__popcll(act));
    } else if (active) {
        atomicAdd(&hist[bucket], 1u);
    }
}

// sorts sel[0, P) ascending (P a power of two, P <= E * TOPK_THREADS, and P == E * TOPK_THREADS when E > 1); all threads of the
// workgroup call it; sel is complete (barrier) before and after
template <int E>
__device__ __forceinline__ void topk_bitonic_regs(unsigned long long *sel, int P) {
    const int tid = threadIdx.x;
    const bool on = tid * E < P;
    unsigned long long v[E];
#pragma unroll
    for (int j = 0; j < E; ++j) v[j] = on ? sel[tid * E + j] : ~0ull;
    for (int size = 2; size <= P; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= 64 * E) {
                __syncthreads();   // (the previous LDS stage's reads are done)
                if (on) {
#pragma unroll
                    for (int j = 0; j < E; ++j) sel[tid * E + j] = v[j];
                }
                __syncthreads();
                if (on) {
#pragma unroll
                    for (int j = 0; j < E; ++j) {
                        const int g = tid * E + j;
                        const unsigned long long b = sel[g ^ stride];
                        const bool up = (g & size) == 0, low = (g & stride) == 0;
                        v[j] = ((v[j] < b) == (low == up)) ? v[j] : b;
                    }
                }
            } else if (stride >= E) {
                const int lm = stride / E;   // partner lane = lane ^ lm, same register
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    const unsigned long long b = __shfl_xor(v[j], lm, 64);
                    const int g = tid * E + j;
                    const bool up = (g & size) == 0, low = (g & stride) == 0;
                    v[j] = ((v[j] < b) == (low == up)) ? v[j] : b;
                }
            } else {   // partners are two of this thread's registers
#pragma unroll
                for (int j = 0; j < E; ++j) {
                    if ((j & stride) == 0 && (j | stride) < E) {
                        const int g = tid * E + j;
                        const bool up = (g & size) == 0;
                        const unsigned long long a = v[j], b = v[j | stride];
                        const bool swap = (a > b) == up;
                        v[j] = swap ? b : a;
                        v[j | stride] = swap ? a : b;
                    }
                }
            }
        }
    }
    __syncthreads();
    if (on) {
#pragma unroll
        for (int j = 0; j < E; ++j) sel[tid * E + j] = v[j];
    }
}

// P = 4096 keys, sorted ascending by 256 threads x 16 keys in registers.  The 12 index bits are split 4 | 4 | 4: in layout A a thread holds
// bits 0-3 (16 consecutive keys), in B bits 4-7, in C bits 8-11, so EVERY compare-exchange stage is register-to-register (about 5
// instructions per pair against about 30 per pair for a shuffled stage); between layouts the keys are transposed through LDS (one pad word
// per 16 keys: 2-way conflicts at worst), 20 transposes in all.  r04, YOLO-sized segments: the sort fell from 34 to 12 us.
// sel holds the keys in plain order on entry and on exit; it must have TOPK_SEL_SIZE entries.  All threads of the workgroup call it.
constexpr int TOPK_SEL_SIZE = TOPK_MAXK + TOPK_MAXK / 16;
template <int SHIFT>
__device__ __forceinline__ void topk_blocked_stages(unsigned long long (&v)[16], int gbase, int size) {
#pragma unroll
    for (int rb = 3; rb >= 0; --rb) {
        const int stride = 1 << (SHIFT + rb);
        if (stride < size) {
            constexpr int dummy = 0; (void)dummy;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if ((j & (1 << rb)) == 0) {
                    const int g = gbase + (j << SHIFT);
                    const bool up = (g & size) == 0;
                    const unsigned long long a = v[j], b = v[j | (1 << rb)];
                    const bool swap = (a > b) == up;
                    v[j] = swap ? b : a;
                    v[j | (1 << rb)] = swap ? a : b;
                }
            }
        }
    }
}
__device__ __forceinline__ void topk_bitonic_blocked_4096(unsigned long long *sel) {
    const int tid = threadIdx.x;
    const bool on = tid < 256;
    const int t = tid & 255, lo = t & 15, hi = t >> 4;
    unsigned long long v[16];
    auto pad = [](int g) { return g + (g >> 4); };
    // index of register j in the three layouts
    auto gA = [&](int j) { return t * 16 + j; };
    auto gB = [&](int j) { return hi * 256 + j * 16 + lo; };
    auto gC = [&](int j) { return j * 256 + t; };
    if (on) {
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = sel[gA(j)];
    }
    int layout = 0;   // 0 = A, 1 = B, 2 = C (uniform)
    auto to_layout = [&](int want) __attribute__((always_inline)) {
        if (want == layout) return;
        __syncthreads();   // (the plain-order reads above / the previous transpose's reads are done)
        if (on) {
#pragma unroll
            for (int j = 0; j < 16; ++j) sel[pad(layout == 0 ? gA(j) : (layout == 1 ? gB(j) : gC(j)))] = v[j];
        }
        __syncthreads();
        if (on) {
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = sel[pad(want == 0 ? gA(j) : (want == 1 ? gB(j) : gC(j)))];
        }
        layout = want;
    };
    for (int size = 2; size <= 4096; size <<= 1) {
        if (size > 256) { to_layout(2); if (on) topk_blocked_stages<8>(v, t, size); }
        if (size > 16) { to_layout(1); if (on) topk_blocked_stages<4>(v, hi * 256 + lo, size); }
        to_layout(0);
        if (on) topk_blocked_stages<0>(v, t * 16, size);
    }
    __syncthreads();
    if (on) {
#pragma unroll
        for (int j = 0; j < 16; ++j) sel[gA(j)] = v[j];
    }
}

// One workgroup per segment.  Selects the k largest scores strictly greater than min_score
// (ties resolved towards the LOWER index), returns them sorted descending (stable).
// seg_off[L+1] (int32, elements).  out_val/out_idx are [L,k]; padded with (-FLT_MAX, 0); out_cnt[L].
// LK: the segment's ordinal keys are staged in LDS first (`lkeys`, n entries) and every later pass reads them there: the passes are
// chains of dependent loads, and from L2 each link costs a round trip (r04: YOLOv5s' 25 200 scores per image).
template <bool LK = false>
__device__ void topk_block_select(const float *__restrict__ scores, const int *__restrict__ seg_off, int seg, int k,
                                  float min_score, float *__restrict__ out_val, int *__restrict__ out_idx,
                                  int *__restrict__ out_cnt, unsigned long long *sel /* LDS, TOPK_SEL_SIZE entries */,
                                  unsigned *lkeys = nullptr) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_remaining, s_count, s_wave_base[TOPK_THREADS / 64], s_tie_taken, s_tie_total;
    const int tid = threadIdx.x;
    const int beg = seg_off[seg], n = seg_off[seg + 1] - beg;
    const float *sc = scores + beg;
    const unsigned omin = ford(min_score);
    TK_STAMP(0);
    // LK: this thread's keys (elements tid, tid + 1024, ...) also stay in REGISTERS: the sweeps below -- count, four radix passes, compaction --
    // then cost a handful of instructions per element (r04: 25 loop iterations of ~100 instructions per sweep were most of the kernel)
    constexpr int PER = LK ? (TOPK_LDS_MAX_N + TOPK_THREADS - 1) / TOPK_THREADS : 1;
    unsigned rk[PER];
    if constexpr (LK) {
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * TOPK_THREADS;
            rk[j] = i < n ? ford(sc[i]) : 0u;   // independent loads: all in flight together
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int i = tid + j * TOPK_THREADS;
            if (i < n) lkeys[i] = rk[j];        // (the LDS copy serves the ordered tie pass, which stops early)
        }
        __syncthreads();
    }
    auto key_at = [&](int i) __attribute__((always_inline)) { return LK ? lkeys[i] : ford(sc[i]); };
    // one sweep over the segment with a uniform trip count (ballots inside the bodies): body(index, key or 0 beyond the end)
    auto sweep = [&](auto body) __attribute__((always_inline)) {
        if constexpr (LK) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                if (j * TOPK_THREADS >= n) break;
                body(j * TOPK_THREADS + tid, rk[j]);
            }
        } else {
            for (int i0 = 0; i0 < n; i0 += TOPK_THREADS) {
                const int i = i0 + tid;
                body(i, i < n ? ford(sc[i]) : 0u);
            }
        }
    };

    TK_STAMP(1);
    // count selectable elements
    if (tid == 0) s_count = 0;
    __syncthreads();
    {
        unsigned c = 0;
        sweep([&](int i, unsigned u) { c += (i < n && u > omin) ? 1u : 0u; });
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((tid & 63) == 0 && c) atomicAdd(&s_count, c);
    }
    __syncthreads();
    const int avail = (int)s_count;
    const int kk = min(k, avail);
    int P = 1;
    while (P < kk) P <<= 1;
    for (int i = tid; i < P; i += TOPK_THREADS) sel[i] = ~0ull;

    TK_STAMP(2);
    unsigned prefix = 0, maskbits = 0, remaining = (unsigned)kk;
    if (kk > 0 && kk < avail) {
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += TOPK_THREADS) hist[i] = 0;
            __syncthreads();
            sweep([&](int i, unsigned u) {
                const bool act = i < n && u > omin && (u & maskbits) == prefix;
                hist_add_aggregated(hist, act, (u >> shift) & 255u);
            });
            __syncthreads();
            // the digit's bin: the largest b with count(bins >= b) >= remaining.  Threads 0..255 own one bin each and take the suffix sums with
            // wave shuffles (r04: one thread walking the 256 bins paid an LDS round trip per bin, four times per segment: a third of the kernel)
            unsigned h_bin = 0, suf = 0;
            if (tid < 256) {
                h_bin = hist[tid];
                suf = h_bin;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const unsigned t = __shfl_down(suf, o, 64);
                    if ((tid & 63) + o < 64) suf += t;
                }
                if ((tid & 63) == 0) s_wave_base[tid >> 6] = suf;
            }
            __syncthreads();
            if (tid < 256) {
                unsigned s_incl = suf;
                for (int w = (tid >> 6) + 1; w < 4; ++w) s_incl += s_wave_base[w];
                const unsigned above = s_incl - h_bin, rem = remaining;
                // (bin 0 also takes the case no bin reaches `remaining`, as the serial walk did)
                if ((s_incl >= rem && above < rem) || (tid == 0 && s_incl < rem)) {
                    s_prefix = prefix | ((unsigned)tid << shift);
                    s_remaining = rem - above;
                }
            }
            __syncthreads();
            prefix = s_prefix;
            remaining = s_remaining;
            maskbits |= 255u << shift;
            __syncthreads();
        }
    } else {
        // take everything selectable: threshold below every selectable key
        prefix = omin;       // u > omin all taken via the "greater" path
        remaining = 0;
    }
    const unsigned thr = prefix;  // exact ordinal of the kk-th largest (or omin)
    TK_STAMP(3);

    // compaction, part 1: strictly-greater elements in any order; count the copies of the k-th value
    if (tid == 0) { s_count = 0; s_tie_taken = 0; s_tie_total = 0; }
    __syncthreads();
    {
        unsigned ties = 0;
        sweep([&](int i, unsigned u) {
            // one LDS atomic per wave reserves the wave's slots (r04: one same-address atomic per selected element serialised -- 4096 of them)
            const bool take = i < n && u > omin && u > thr;
            const unsigned long long bal = __ballot(take);
            if (bal != 0ull) {
                const int ln = tid & 63, leader = __ffsll((long long)bal) - 1;
                unsigned base_pos = 0;
                if (ln == leader) base_pos = atomicAdd(&s_count, (unsigned)__popcll(bal));
                base_pos = __builtin_amdgcn_readlane(base_pos, leader);
                if (take) {
                    const unsigned pos = base_pos + (unsigned)__popcll(bal & ((1ull << ln) - 1ull));
                    if (pos < (unsigned)TOPK_MAXK) sel[pos] = ((unsigned long long)(~u) << 32) | (unsigned)i;
                }
            }
            ties += (i < n && u > omin && u == thr && remaining > 0) ? 1u : 0u;
        });
        for (int o = 32; o > 0; o >>= 1) ties += __shfl_xor(ties, o, 64);
        if ((tid & 63) == 0 && ties) atomicAdd(&s_tie_total, ties);
    }
    __syncthreads();
    // part 2: copies of the k-th value.  When exactly the missing number exists they are all taken, in any order (the sort below
    // orders by (value, index)): no barriers.  r01: the barrier-per-1024-elements ordered scan used to run for every segment --
    // 48 barriers on the 16 384-element CenterNet maps.  It is needed only when the k-th value has more copies than are missing.
    if (remaining > 0 && s_tie_total == remaining) {
        sweep([&](int i, unsigned u) {
            if (i < n && u > omin && u == thr) {
                const unsigned pos = (unsigned)kk - remaining + atomicAdd(&s_tie_taken, 1u);
                sel[pos] = ((unsigned long long)(~u) << 32) | (unsigned)i;
            }
        });
    } else if (remaining > 0) {
        // ordered (index-ascending) selection of the first `remaining` copies; stops (uniformly) once they are found
        for (int i0 = 0; i0 < n; i0 += TOPK_THREADS) {
            const int i = i0 + tid;
            const unsigned u = i < n ? key_at(i) : 0u;
            const bool tie = i < n && u > omin && u == thr;
            const unsigned long long bal = __ballot(tie);
            const int wv = tid >> 6, ln = tid & 63;
            if (ln == 0) s_wave_base[wv] = (unsigned)__popcll(bal);
            __syncthreads();
            if (tid == 0) {
                unsigned run = s_tie_taken;
                for (int w = 0; w < TOPK_THREADS / 64; ++w) { const unsigned c = s_wave_base[w]; s_wave_base[w] = run; run += c; }
                s_tie_taken = run;
            }
            __syncthreads();
            if (tie) {
                const unsigned rank = s_wave_base[wv] + (unsigned)__popcll(bal & ((1ull << ln) - 1ull));
                if (rank < remaining) sel[(unsigned)kk - remaining + rank] = ((unsigned long long)(~u) << 32) | (unsigned)i;
            }
            const bool done = s_tie_taken >= remaining;   // every thread reads the same value between the two barriers
            __syncthreads();
            if (done) break;
        }
    }
    // (greater elements occupy [0, kk-remaining), ties [kk-remaining, kk))
    // bitonic sort ascending on (~key, idx)  == key descending, index ascending
    TK_STAMP(4);
    // bitonic sort, ascending keys.  Thread t keeps entries t E .. t E + E - 1 in registers: a stage whose partners are less than E apart is
    // register-to-register, less than 64 E apart a wave shuffle, and only the farther ones (10 of the 78 stages at P = 4096) go through
    // LDS with workgroup barriers (r04: one barrier + LDS round trip per stage was 33 of the kernel's 73 us on YOLOv5s' segments)
    __syncthreads();
    if (P > 2048) topk_bitonic_blocked_4096(sel);
    else if (P > 1024) topk_bitonic_regs<2>(sel, P);
    else topk_bitonic_regs<1>(sel, P);
    __syncthreads();
    TK_STAMP(5);
    for (int i = tid; i < k; i += TOPK_THREADS) {
        float v = -FLT_MAX;
        int id = 0;
        if (i < kk) {
            const unsigned long long e = sel[i];
            v = funord(~(unsigned)(e >> 32));
            id = (int)(unsigned)(e & 0xffffffffu);
        }
        out_val[(size_t)seg * k + i] = v;
        out_idx[(size_t)seg * k + i] = id;
    }
    if (tid == 0) out_cnt[seg] = kk;
    TK_STAMP(6);
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_segmented_kernel(const float *__restrict__ scores,
                                                                      const int *__restrict__ seg_off, int k,
                                                                      float min_score, float *__restrict__ out_val,
                                                                      int *__restrict__ out_idx,
                                                                      int *__restrict__ out_cnt) {
    __shared__ unsigned long long sel[TOPK_SEL_SIZE];
    topk_block_select(scores, seg_off, blockIdx.x, k, min_score, out_val, out_idx, out_cnt, sel);
}

// segments of at most `cap` (<= TOPK_LDS_MAX_N) scores: keys staged in dynamic LDS; a longer segment (a caller's max_segment that was
// not a bound after all) takes the global-memory passes
__global__ __launch_bounds__(TOPK_THREADS) void topk_segmented_lds_kernel(const float *__restrict__ scores, const int *__restrict__ seg_off, int k,
                                                                          float min_score, float *__restrict__ out_val, int *__restrict__ out_idx,
                                                                          int *__restrict__ out_cnt, int cap) {
    __shared__ unsigned long long sel[TOPK_SEL_SIZE];
    extern __shared__ __attribute__((aligned(16))) unsigned topk_lkeys[];
    const int n = seg_off[blockIdx.x + 1] - seg_off[blockIdx.x];
    if (n <= cap) topk_block_select<true>(scores, seg_off, blockIdx.x, k, min_score, out_val, out_idx, out_cnt, sel, topk_lkeys);
    else topk_block_select<false>(scores, seg_off, blockIdx.x, k, min_score, out_val, out_idx, out_cnt, sel);
}

// ---- multi-workgroup path for long segments (RPN level P2: 201 600 scores per image).
// A: per-chunk LDS histogram of the top 11 ordinal bits -> global histogram per segment.
// C: every chunk finds the boundary bin b1 (count(bin > b1) < k <= count(bin >= b1)) and appends the
//    candidates (bin >= b1) as 64-bit keys (~ordinal << 32 | index) to the segment's candidate list.
// D: one workgroup per segment sorts the (few thousand) candidates in LDS: ascending key order ==
//    descending score, ascending index -> exact stable top-k.  If the candidate list overflowed (extreme
//    ties) the segment falls back to the single-workgroup select above.
constexpr int TK_BINS = 2048;
constexpr int TK_CHUNK = 8192;
constexpr int TK_CAP = 8192;

__global__ __launch_bounds__(256) void topk_hist_kernel(const float *__restrict__ scores, const int *__restrict__ seg_off,
                                                        float min_score, unsigned *__restrict__ ghist) {
    __shared__ unsigned hist[TK_BINS];
    const int seg = blockIdx.y, tid = threadIdx.x;
    const int beg = seg_off[seg], n = seg_off[seg + 1] - beg;
    const int c0 = blockIdx.x * TK_CHUNK;
    if (c0 >= n) return;
    const int c1 = min(n, c0 + TK_CHUNK);
    for (int i = tid; i < TK_BINS; i += 256) hist[i] = 0;
    __syncthreads();
    const unsigned omin = ford(min_score);
    for (int i = c0 + tid; i < c1; i += 256) {
        const unsigned u = ford(scores[beg + i]);
        if (u > omin) atomicAdd(&hist[u >> 21], 1u);
    }
    __syncthreads();
    for (int i = tid; i < TK_BINS; i += 256) {
        const unsigned v = hist[i];
        if (v) atomicAdd(&ghist[(size_t)seg * TK_BINS + i], v);
    }
}

__global__ __launch_bounds__(256) void topk_compact_kernel(const float *__restrict__ scores, const int *__restrict__ seg_off,
                                                           int k, float min_score, const unsigned *__restrict__ ghist,
                                                           unsigned *__restrict__ cand_cnt,
                                                           unsigned long long *__restrict__ cand) {
    __shared__ unsigned part[256];
    __shared__ int s_b1;
    const int seg = blockIdx.y, tid = threadIdx.x;
    const int beg = seg_off[seg], n = seg_off[seg + 1] - beg;
    const int c0 = blockIdx.x * TK_CHUNK;
    if (c0 >= n) return;
    const int c1 = min(n, c0 + TK_CHUNK);
    // boundary bin: thread t owns bins [8t, 8t+8); suffix sums from the top via a wave scan + 4 wave totals
    const unsigned *gh = ghist + (size_t)seg * TK_BINS;
    unsigned mine[8], tot = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) { mine[j] = gh[tid * 8 + j]; tot += mine[j]; }
    const int lane = tid & 63, wv = tid >> 6;
    unsigned suf = tot;  // inclusive suffix sum inside the wave (lanes >= lane)
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_down(suf, o, 64);
        if (lane + o < 64) suf += t;
    }
    if (lane == 0) part[wv] = suf;  // wave total
    if (tid == 0) s_b1 = 0;
    __syncthreads();
    unsigned above = suf - tot;  // bins owned by higher lanes of this wave
    for (int w = wv + 1; w < 4; ++w) above += part[w];
    if (above < (unsigned)k && above + tot >= (unsigned)k) {
        unsigned c = above;
        int b = tid * 8 + 7;
        for (int j = 7; j >= 0; --j) {
            if (c + mine[j] >= (unsigned)k) { b = tid * 8 + j; break; }
            c += mine[j];
        }
        s_b1 = b;
    }
    __syncthreads();
    const unsigned b1 = (unsigned)s_b1;  // stays 0 when fewer than k scores are selectable: everything is a candidate
    const unsigned omin = ford(min_score);
    // The chunk's 32 elements per thread stay in registers; ONE atomicAdd per workgroup reserves its slice of the
    // candidate list (r01: one same-address atomic per candidate serialised in L2 and made this kernel 10x the
    // histogram pass).  Candidate order is irrelevant: topk_final_kernel sorts by (score, index).
    constexpr int PER = TK_CHUNK / 256;
    unsigned u[PER];
    unsigned cnt = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = c0 + tid + 256 * j;
        u[j] = i < c1 ? ford(scores[beg + i]) : 0u;
        if (i >= c1 || !(u[j] > omin && (u[j] >> 21) >= b1)) u[j] = 0u;  // ford() of a selectable score is never 0
        cnt += u[j] != 0u;
    }
    unsigned inc = cnt;  // inclusive prefix over the wave, then wave bases through LDS
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();  // part[] is reused
    if (lane == 63) part[wv] = inc;
    __syncthreads();
    unsigned wbase = 0, total = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wv) wbase += part[w];
        total += part[w];
    }
    if (tid == 0) part[8] = total ? atomicAdd(&cand_cnt[seg], total) : 0u;
    __syncthreads();
    unsigned pos = part[8] + wbase + inc - cnt;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        if (u[j] != 0u) {
            if (pos < (unsigned)TK_CAP) cand[(size_t)seg * TK_CAP + pos] = ((unsigned long long)(~u[j]) << 32) | (unsigned)(c0 + tid + 256 * j);
            ++pos;
        }
    }
}

__global__ __launch_bounds__(TOPK_THREADS) void topk_final_kernel(const float *__restrict__ scores,
                                                                  const int *__restrict__ seg_off, int k, float min_score,
                                                                  const unsigned *__restrict__ cand_cnt,
                                                                  const unsigned long long *__restrict__ cand,
                                                                  float *__restrict__ out_val, int *__restrict__ out_idx,
                                                                  int *__restrict__ out_cnt) {
    __shared__ unsigned long long keys[TK_CAP];
    const int seg = blockIdx.x, tid = threadIdx.x;
    const unsigned nc = cand_cnt[seg];
    if (nc > (unsigned)TK_CAP) {  // block-uniform
        topk_block_select(scores, seg_off, seg, k, min_score, out_val, out_idx, out_cnt, keys);
        return;
    }
    int P = 1;
    while (P < (int)nc) P <<= 1;
    for (int i = tid; i < P; i += TOPK_THREADS) keys[i] = i < (int)nc ? cand[(size_t)seg * TK_CAP + i] : ~0ull;
    __syncthreads();
    // (r04: the register / shuffle forms of the single-workgroup select instead of one barrier + LDS round trip per stage)
    if (P > 4096) topk_bitonic_regs<8>(keys, P);
    else if (P > 2048) topk_bitonic_blocked_4096(keys);
    else if (P > 1024) topk_bitonic_regs<2>(keys, P);
    else topk_bitonic_regs<1>(keys, P);
    __syncthreads();
    const int kk = min(k, (int)nc);
    for (int i = tid; i < k; i += TOPK_THREADS) {
        float v = -FLT_MAX;
        int id = 0;
        if (i < kk) {
            const unsigned long long e = keys[i];
            v = funord(~(unsigned)(e >> 32));
            id = (int)(unsigned)(e & 0xffffffffu);
        }
        out_val[(size_t)seg * k + i] = v;
        out_idx[(size_t)seg * k + i] = id;
    }
    if (tid == 0) out_cnt[seg] = kk;
}

// ------------------------------------------------------------------------------------------ RoIAlign (FPN, NHWC bf16)
struct RoiLevel { const uint16_t *feat; int H, W; float scale; };
struct RoiArgs {
    RoiLevel lv[6];
    int L, C, P, sampling, aligned;
    int k_min, canonical_level; float canonical_scale;
    int N;  // batch images
    // level map by exact comparisons: RoI goes to level k_min + #{j in 1..L-1 : area >= lvl_thr[j-1]}, where
    // lvl_thr[j-1] = fp32((canonical_scale * (2^(k_min+j-canonical_level) - 1e-6))^2) (computed in double on the host): the
    // same predicate as floor(k0 + log2(sqrt(wh)/224 + 1e-6)) >= k_min + j (Lin et al. 2017 eq. 1) without a transcendental,
    // so the integer level is bit-exact against any restatement that compares the fp32 area with the same table
    float lvl_thr[5];
};
__device__ __forceinline__ int roi_fpn_level(const RoiArgs &a, const float *roi) {
    const float w = roi[3] - roi[1], h = roi[4] - roi[2];
    const float area = fmaxf(__fmul_rn(w, h), 0.f);
    int lvl = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j) lvl += (j < a.L - 1 && area >= a.lvl_thr[j]) ? 1 : 0;
    return lvl;
}
__device__ __forceinline__ float rbf2f(unsigned v16) { return __uint_as_float(v16 << 16); }
__device__ __forceinline__ unsigned rf2bf(float f) {
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40;
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}
// rois [R,5] = (batch_idx, x1,y1,x2,y2) f32 ; out [R,P,P,C] bf16 ; one thread = 8 channels of one bin
// two fp32 -> packed bf16, round-to-nearest-even, one instruction (gfx950)
__device__ __forceinline__ unsigned rpk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__global__ void roi_align_kernel(RoiArgs a, const float *__restrict__ rois, int R, uint16_t *__restrict__ out,
                                 int *__restrict__ out_level) {
    const int cv = a.C / 8;
    const size_t total = (size_t)R * a.P * a.P * cv;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(e % cv);
        size_t t = e / cv;
        const int pw = (int)(t % a.P); t /= a.P;
        const int ph = (int)(t % a.P);
        const int r = (int)(t / a.P);
        const float *roi = rois + (size_t)r * 5;
        const int b = (int)roi[0];
        const int lvl = roi_fpn_level(a, roi);
        if (out_level && c8 == 0 && ph == 0 && pw == 0) out_level[r] = lvl + a.k_min;
        const RoiLevel L = a.lv[lvl];
        const float off = a.aligned ? 0.5f : 0.f;
        const float x1 = roi[1] * L.scale - off, y1 = roi[2] * L.scale - off;
        const float x2 = roi[3] * L.scale - off, y2 = roi[4] * L.scale - off;
        float rw = x2 - x1, rh = y2 - y1;
        if (!a.aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
        const float bw = rw / (float)a.P, bh = rh / (float)a.P;
        const int g = a.sampling;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const uint16_t *base = L.feat + (size_t)b * L.H * L.W * a.C + c8 * 8;
        for (int iy = 0; iy < g; ++iy) {
            const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)g;
            for (int ix = 0; ix < g; ++ix) {
                const float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)g;
                if (y < -1.0f || y > (float)L.H || x < -1.0f || x > (float)L.W) continue;
                float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
                int y_lo = (int)yy, x_lo = (int)xx, y_hi, x_hi;
                if (y_lo >= L.H - 1) { y_hi = y_lo = L.H - 1; yy = (float)y_lo; } else y_hi = y_lo + 1;
                if (x_lo >= L.W - 1) { x_hi = x_lo = L.W - 1; xx = (float)x_lo; } else x_hi = x_lo + 1;
                const float ly = yy - (float)y_lo, lx = xx - (float)x_lo, hy = 1.f - ly, hx = 1.f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const uint4 v1 = *reinterpret_cast<const uint4 *>(base + ((size_t)y_lo * L.W + x_lo) * a.C);
                const uint4 v2 = *reinterpret_cast<const uint4 *>(base + ((size_t)y_lo * L.W + x_hi) * a.C);
                const uint4 v3 = *reinterpret_cast<const uint4 *>(base + ((size_t)y_hi * L.W + x_lo) * a.C);
                const uint4 v4 = *reinterpret_cast<const uint4 *>(base + ((size_t)y_hi * L.W + x_hi) * a.C);
                const unsigned q1[4] = {v1.x, v1.y, v1.z, v1.w}, q2[4] = {v2.x, v2.y, v2.z, v2.w};
                const unsigned q3[4] = {v3.x, v3.y, v3.z, v3.w}, q4[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[2 * q] += w1 * rbf2f(q1[q] & 0xffffu) + w2 * rbf2f(q2[q] & 0xffffu) + w3 * rbf2f(q3[q] & 0xffffu) +
                                  w4 * rbf2f(q4[q] & 0xffffu);
                    acc[2 * q + 1] += w1 * rbf2f(q1[q] >> 16) + w2 * rbf2f(q2[q] >> 16) + w3 * rbf2f(q3[q] >> 16) +
                                      w4 * rbf2f(q4[q] >> 16);
                }
            }
        }
        const float inv = 1.f / (float)(g * g);
        uint4 o;
        o.x = rpk_bf16(acc[0] * inv, acc[1] * inv);
        o.y = rpk_bf16(acc[2] * inv, acc[3] * inv);
        o.z = rpk_bf16(acc[4] * inv, acc[5] * inv);
        o.w = rpk_bf16(acc[6] * inv, acc[7] * inv);
        *reinterpret_cast<uint4 *>(out + e * 8) = o;
    }
}

// 32 channels per thread (C % 32 == 0): the RoI / bin / tap arithmetic is paid once per 64 B instead of once per 16 B, and
// the accumulation runs on channel PAIRS (v_pk_fma_f32).  Same sampling arithmetic as roi_align_kernel; the 16 tap terms of
// a bin are fused-multiply-added one by one (different fp32 rounding than the 4-term sums there, same bf16 tolerance).  r01: the 16-B kernel was VALU-bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void roi_align_c32_kernel(RoiArgs a, const float *__restrict__ rois, int R, uint16_t *__restrict__ out,
                                                            int *__restrict__ out_level) {
    const int cv = a.C / 32;
    const size_t total = (size_t)R * a.P * a.P * cv;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int c32 = (int)(e % cv);
        size_t t = e / cv;
        const int pw = (int)(t % a.P); t /= a.P;
        const int ph = (int)(t % a.P);
        const int r = (int)(t / a.P);
        const float *roi = rois + (size_t)r * 5;
        const int b = (int)roi[0];
        const int lvl = roi_fpn_level(a, roi);
        if (out_level && c32 == 0 && ph == 0 && pw == 0) out_level[r] = lvl + a.k_min;
        const RoiLevel L = a.lv[lvl];
        const float off = a.aligned ? 0.5f : 0.f;
        const float x1 = roi[1] * L.scale - off, y1 = roi[2] * L.scale - off;
        const float x2 = roi[3] * L.scale - off, y2 = roi[4] * L.scale - off;
        float rw = x2 - x1, rh = y2 - y1;
        if (!a.aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
        const float bw = rw / (float)a.P, bh = rh / (float)a.P;
        const int g = a.sampling;
        f32x2 acc[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = (f32x2){0.f, 0.f};
        // the thread's four 16-B chunks are cv chunks apart (chunks c32, c32 + cv, ...): a load instruction of the cv threads of a
        // bin then covers one contiguous cv*16-B run of the pixel (whole 128-B lines for C = 256) instead of every fourth chunk
        const uint16_t *base = L.feat + (size_t)b * L.H * L.W * a.C + c32 * 8;
        if (g == 2) {
            // sampling_ratio 2 (the detectors' setting): the 2 x 2 samples of a bin form a grid, so the weight of pixel (r, c) is
            // (sum over sample rows of the row's weight at r) x (the same over sample columns at c), and neighbouring samples share rows /
            // columns whenever the bin is under 2 pixels wide: the up to 4 rows and 4 columns are DEDUPLICATED and each distinct pixel is
            // loaded once -- 4 to 16 pixel loads per bin instead of always 16 (r03: the kernel is bound by the bytes through the CU's
            // vector L1, 16 taps x 512 B per bin; typical RoIs at their FPN level have 1-4-pixel bins).  fp32 sums in another order than the
            // tap-by-tap form: same bf16 tolerance.
            int ri[4], ci[4];
            float wy[4], wx[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float y = y1 + (float)ph * bh + ((float)i + 0.5f) * bh * 0.5f;
                const float x = x1 + (float)pw * bw + ((float)i + 0.5f) * bw * 0.5f;
                const bool oky = !(y < -1.0f || y > (float)L.H), okx = !(x < -1.0f || x > (float)L.W);
                float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
                int y_lo = (int)yy, x_lo = (int)xx, y_hi, x_hi;
                if (y_lo >= L.H - 1) { y_hi = y_lo = L.H - 1; yy = (float)y_lo; } else y_hi = y_lo + 1;
                if (x_lo >= L.W - 1) { x_hi = x_lo = L.W - 1; xx = (float)x_lo; } else x_hi = x_lo + 1;
                const float ly = yy - (float)y_lo, lx = xx - (float)x_lo;
                ri[2 * i] = y_lo; ri[2 * i + 1] = y_hi; wy[2 * i] = oky ? 1.f - ly : 0.f; wy[2 * i + 1] = oky ? ly : 0.f;
                ci[2 * i] = x_lo; ci[2 * i + 1] = x_hi; wx[2 * i] = okx ? 1.f - lx : 0.f; wx[2 * i + 1] = okx ? lx : 0.f;
            }
            // merge equal indices into the earliest slot (the later slot's weight becomes 0 and its pixel is never loaded)
#pragma unroll
            for (int j = 1; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < j; ++k) {
                    if (ri[j] == ri[k] && wy[j] != 0.f) { wy[k] += wy[j]; wy[j] = 0.f; }
                    if (ci[j] == ci[k] && wx[j] != 0.f) { wx[k] += wx[j]; wx[j] = 0.f; }
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (wy[i] == 0.f) continue;
                const uint16_t *rowp = base + (size_t)ri[i] * L.W * a.C;
                // (r03, tried: requesting the row's four pixels together before accumulating -- 16 loads in flight per lane -- costs 170 instead of ~100
                // registers and the occupancy it loses: 0.96 -> 1.16 ms per 60 images.  The kernel moves 7.7 GB through HBM per 120 images, ~1.1 x compulsory.)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float w = wy[i] * wx[j];
                    if (w == 0.f) continue;
                    const uint4 *pp = reinterpret_cast<const uint4 *>(rowp + (size_t)ci[j] * a.C);
                    uint4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = pp[q * cv];
                    const f32x2 w2 = (f32x2){w, w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned wd[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const f32x2 pr = (f32x2){__uint_as_float(wd[d] << 16), __uint_as_float(wd[d] & 0xffff0000u)};
                            acc[q * 4 + d] = __builtin_elementwise_fma(w2, pr, acc[q * 4 + d]);
                        }
                    }
                }
            }
        } else
        for (int iy = 0; iy < g; ++iy) {
            const float y = y1 + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)g;
            for (int ix = 0; ix < g; ++ix) {
                const float x = x1 + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)g;
                if (y < -1.0f || y > (float)L.H || x < -1.0f || x > (float)L.W) continue;
                float yy = fmaxf(y, 0.f), xx = fmaxf(x, 0.f);
                int y_lo = (int)yy, x_lo = (int)xx, y_hi, x_hi;
                if (y_lo >= L.H - 1) { y_hi = y_lo = L.H - 1; yy = (float)y_lo; } else y_hi = y_lo + 1;
                if (x_lo >= L.W - 1) { x_hi = x_lo = L.W - 1; xx = (float)x_lo; } else x_hi = x_lo + 1;
                const float ly = yy - (float)y_lo, lx = xx - (float)x_lo, hy = 1.f - ly, hx = 1.f - lx;
                const float wt[4] = {hy * hx, hy * lx, ly * hx, ly * lx};
                const uint16_t *ptr[4] = {base + ((size_t)y_lo * L.W + x_lo) * a.C, base + ((size_t)y_lo * L.W + x_hi) * a.C,
                                          base + ((size_t)y_hi * L.W + x_lo) * a.C, base + ((size_t)y_hi * L.W + x_hi) * a.C};
#pragma unroll
                for (int tp = 0; tp < 4; ++tp) {
                    uint4 v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const uint4 *>(ptr[tp])[q * cv];
                    const f32x2 w2 = (f32x2){wt[tp], wt[tp]};
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned wd[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            const f32x2 pr = (f32x2){__uint_as_float(wd[d] << 16), __uint_as_float(wd[d] & 0xffff0000u)};
                            acc[q * 4 + d] = __builtin_elementwise_fma(w2, pr, acc[q * 4 + d]);
                        }
                    }
                }
            }
        }
        const float inv = 1.f / (float)(g * g);
        uint4 *dst = reinterpret_cast<uint4 *>(out + (e / cv) * a.C + c32 * 8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 o;
            o.x = rpk_bf16(acc[q * 4 + 0].x * inv, acc[q * 4 + 0].y * inv);
            o.y = rpk_bf16(acc[q * 4 + 1].x * inv, acc[q * 4 + 1].y * inv);
            o.z = rpk_bf16(acc[q * 4 + 2].x * inv, acc[q * 4 + 2].y * inv);
            o.w = rpk_bf16(acc[q * 4 + 3].x * inv, acc[q * 4 + 3].y * inv);
            // non-temporal: the 3 GB pooled tensor is not read again by this kernel and should not push feature-map lines out of L2
            __builtin_nontemporal_store((u32x4_t){o.x, o.y, o.z, o.w}, reinterpret_cast<u32x4_t *>(dst + q * cv));
        }
    }
}

// ------------------------------------------------------------------------------------------ CenterNet decode
// heat [B,C,H,W] f32 (already sigmoid+clip).  keep = (heat == maxpool3x3_same(heat)); out = heat*keep.
__global__ void heat_nms_kernel(const float *__restrict__ heat, float *__restrict__ out, int H, int W, size_t planes) {
    const size_t total = planes * H * W;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(e % W);
        const int y = (int)((e / W) % H);
        const float *p = heat + (e - (size_t)y * W - x);
        const float v = heat[e];
        float m = v;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int dx = -1; dx <= 1; ++dx) {
                const int xx = x + dx;
                if ((unsigned)xx >= (unsigned)W) continue;
                m = fmaxf(m, p[(size_t)yy * W + xx]);
            }
        }
        out[e] = v == m ? v : v * 0.0f;
    }
}
// The three passes NHWC bf16 head -> NCHW fp32 (md_nhwc_to_nchw_f32), sigmoid + clip (md_sigmoid_clip), 3x3 peak test (md_heat_nms) in
// ONE kernel with the same arithmetic per element: a workgroup owns an 8 x 64 pixel tile of 16 classes of one image, stages the
// sigmoid-clipped values of the tile + a 1-pixel halo in LDS (out-of-image = -FLT_MAX: ignored by the max, like the bounds checks
// of heat_nms_kernel) and writes the NCHW planes with 256-B rows.  r01: the three passes were 0.58 ms of the 3.46 ms CenterNet step.
struct PeakArgs { int H, W, Cp, c0, nc; float lo, hi; };
constexpr int PK_TH = 8, PK_TW = 64, PK_C = 16;
__global__ __launch_bounds__(256) void heat_peaks_kernel(const uint16_t *__restrict__ head, PeakArgs a, float *__restrict__ heat,
                                                         float *__restrict__ hm) {
    __shared__ float s[PK_C][PK_TH + 2][PK_TW + 3];
    const int tiles_x = (a.W + PK_TW - 1) / PK_TW;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int cg = blockIdx.y, b = blockIdx.z;
    const int y0 = ty * PK_TH - 1, x0 = tx * PK_TW - 1;
    constexpr int HPIX = (PK_TH + 2) * (PK_TW + 2);
    for (int i = threadIdx.x; i < HPIX * 2; i += 256) {
        const int pix = i >> 1, q = i & 1;
        const int hy = pix / (PK_TW + 2), hx = pix - hy * (PK_TW + 2);
        const int y = y0 + hy, x = x0 + hx, ch0 = cg * PK_C + q * 8;
        float v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = -FLT_MAX;
        if ((unsigned)y < (unsigned)a.H && (unsigned)x < (unsigned)a.W && ch0 < a.nc) {
            const uint4 r = *reinterpret_cast<const uint4 *>(head + ((size_t)(b * a.H + y) * a.W + x) * a.Cp + a.c0 + ch0);
            const unsigned w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float xin = __uint_as_float(k & 1 ? (w[k >> 1] & 0xffff0000u) : (w[k >> 1] << 16));
                const float sg = 1.0f / (1.0f + expf(-xin));
                v[k] = fminf(fmaxf(sg, a.lo), a.hi);
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) s[q * 8 + k][hy][hx] = v[k];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < PK_C * PK_TH * PK_TW; e += 256) {
        const int x = e % PK_TW, y = (e / PK_TW) % PK_TH, c = e / (PK_TW * PK_TH);
        const int gy = ty * PK_TH + y, gx = tx * PK_TW + x, ch = cg * PK_C + c;
        if (gy >= a.H || gx >= a.W || ch >= a.nc) continue;
        const float v = s[c][y + 1][x + 1];
        float m = v;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) m = fmaxf(m, s[c][y + dy][x + dx]);
        const size_t o = ((size_t)(b * a.nc + ch) * a.H + gy) * a.W + gx;
        heat[o] = v == m ? v : v * 0.0f;
        if (hm) hm[o] = v;
    }
}

// second stage: per image, from per-class top-K (scores [B,C,K], inds [B,C,K]) the global top-K was
// selected by md_topk_segmented over [B, C*K]; assemble detections.
__global__ void centernet_assemble_kernel(const float *__restrict__ top_score, const int *__restrict__ top_ind2,
                                          const int *__restrict__ cls_inds, const float *__restrict__ wh,
                                          const float *__restrict__ reg, int B, int C, int K, int H, int W,
                                          float *__restrict__ det, int *__restrict__ out_inds, int *__restrict__ out_cls) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * K) return;
    const int b = i / K;
    const int j = top_ind2[i];         // index into [C*K]
    const int cls = j / K;
    const int ind = cls_inds[(size_t)b * C * K + j];
    const float ys0 = (float)(ind / W), xs0 = (float)(ind % W);
    const size_t hw = (size_t)H * W;
    const float w_ = wh[((size_t)b * 2 + 0) * hw + ind], h_ = wh[((size_t)b * 2 + 1) * hw + ind];
    float xs = xs0, ys = ys0;
    if (reg) { xs = xs + reg[((size_t)b * 2 + 0) * hw + ind]; ys = ys + reg[((size_t)b * 2 + 1) * hw + ind]; }
    else { xs = xs + 0.5f; ys = ys + 0.5f; }
    float *d = det + (size_t)i * 6;
    d[0] = xs - w_ / 2; d[1] = ys - h_ / 2; d[2] = xs + w_ / 2; d[3] = ys + h_ / 2;
    d[4] = top_score[i];
    d[5] = (float)cls;
    out_inds[i] = ind;
    out_cls[i] = cls;
}

// ------------------------------------------------------------------------------------------ CenterPoint head
// centerpoint/det3d_ms/models/bbox_heads/center_head.py:297-345 (predict) + :398-430 (post_processing up to
// the TopK): per BEV cell sigmoid(hm) max/argmax, exp(dim), atan2(rot), centre = (cell + reg) * out_size_factor *
// voxel + pc_range; score/range mask -> score = label = -1, box = 0; heading flipped for the NMS operator
// (box[:, -1] = -rot - pi/2, :426) and dims swapped in the NMS copy (:430).
struct CpArgs {
    int o_reg, o_height, o_dim, o_rot, o_vel, o_hm, ncls, C;
    int H, W;
    float score_thr, osf, vx, vy, px, py;
    float rmin[3], rmax[3];
};
__global__ void centerpoint_decode_kernel(const uint16_t *__restrict__ head, CpArgs a, int total,
                                          float *__restrict__ scores, int *__restrict__ labels,
                                          float *__restrict__ boxes, float *__restrict__ nms_boxes) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int loc = e % (a.H * a.W);
    const int y = loc / a.W, x = loc % a.W;
    const uint16_t *h = head + (size_t)e * a.C;
    float best = -FLT_MAX;
    int lab = 0;
    for (int c = 0; c < a.ncls; ++c) {  // ArgMaxWithValue: first maximum wins
        const float v = 1.0f / (1.0f + expf(-rbf2f(h[a.o_hm + c])));
        if (v > best) { best = v; lab = c; }
    }
    const float xs = ((float)x + rbf2f(h[a.o_reg])) * a.osf * a.vx + a.px;
    const float ys = ((float)y + rbf2f(h[a.o_reg + 1])) * a.osf * a.vy + a.py;
    const float zs = rbf2f(h[a.o_height]);
    const float d0 = expf(rbf2f(h[a.o_dim])), d1 = expf(rbf2f(h[a.o_dim + 1])), d2 = expf(rbf2f(h[a.o_dim + 2]));
    const float rot = atan2f(rbf2f(h[a.o_rot]), rbf2f(h[a.o_rot + 1]));
    const float v0 = a.o_vel >= 0 ? rbf2f(h[a.o_vel]) : 0.f, v1 = a.o_vel >= 0 ? rbf2f(h[a.o_vel + 1]) : 0.f;
    const bool in_range = xs >= a.rmin[0] && ys >= a.rmin[1] && zs >= a.rmin[2] && xs <= a.rmax[0] && ys <= a.rmax[1] && zs <= a.rmax[2];
    const bool ok = best > a.score_thr && in_range;
    float *b = boxes + (size_t)e * 9;
    float *nb = nms_boxes + (size_t)e * 7;
    if (ok) {
        b[0] = xs; b[1] = ys; b[2] = zs; b[3] = d0; b[4] = d1; b[5] = d2; b[6] = v0; b[7] = v1; b[8] = rot;
        const float r2 = -rot - 1.5707963267948966f;
        nb[0] = xs; nb[1] = ys; nb[2] = zs; nb[3] = d1; nb[4] = d0; nb[5] = d2; nb[6] = r2;
        scores[e] = best; labels[e] = lab;
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) b[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) nb[k] = 0.f;
        nb[6] = -1.5707963267948966f;  // -0 - pi/2, as the reference's in-place update of a zeroed row gives
        scores[e] = -1.f; labels[e] = -1;
    }
}

// YOLOv5 Detect decode: per (cell, anchor): sigmoid of the 5+nc outputs; xy = (2s - 0.5 + grid) * stride,
// wh = (2s)^2 * anchor; score = obj * max_c cls_c (single-label mode), label = argmax.
struct YoloArgs { int H, W, Cp, nc, A; float stride, aw[3], ah[3], thr; int off, total; };
// Head rows are staged through LDS with coalesced 16-B loads (cells consecutive cells x Cp channels; rows padded by one dword so
// that the per-(cell, anchor) scalar reads that follow spread over the banks): r01 the direct 2-byte global reads at a 170-B lane
// stride made this kernel 24 % of the YOLOv5s step.  The class arg-max runs on the logits (sigmoid is monotonic; logits are
// clamped to the range where the fp32 sigmoid still separates values, so saturated ties resolve to the first index as they
// would after the sigmoid) and only the winner is passed through the sigmoid.
__device__ __forceinline__ void stage_head_rows(const uint16_t *__restrict__ head, unsigned *sm, long long cell0, long long cells_total,
                                                int cells, int Cp) {
    const int chunks = Cp / 8, rowdw = Cp / 2 + 1;
    for (int i = threadIdx.x; i < cells * chunks; i += blockDim.x) {
        const int c = i / chunks, q = i - c * chunks;
        if (cell0 + c < cells_total) {
            const uint4 v = *reinterpret_cast<const uint4 *>(head + (size_t)(cell0 + c) * Cp + q * 8);
            unsigned *d = sm + c * rowdw + q * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
    }
    __syncthreads();
}
__device__ __forceinline__ float logit_key(float v) { return fminf(fmaxf(v, -87.f), 16.f); }

__global__ __launch_bounds__(256) void yolo_decode_kernel(const uint16_t *__restrict__ head, YoloArgs a, int B, int cells,
                                                          float *__restrict__ boxes, float *__restrict__ scores, int *__restrict__ labels) {
    extern __shared__ unsigned ysm[];
    const long long cells_total = (long long)B * a.H * a.W, cell0 = (long long)blockIdx.x * cells;
    stage_head_rows(head, ysm, cell0, cells_total, cells, a.Cp);
    const int t = threadIdx.x;
    if (t >= cells * a.A) return;
    const int c = t / a.A, an = t - c * a.A;
    const long long cell = cell0 + c;
    if (cell >= cells_total) return;
    const int b = (int)(cell / (a.H * a.W)), loc = (int)(cell - (long long)b * a.H * a.W);
    const int gx = loc % a.W, gy = loc / a.W;
    const uint16_t *h = reinterpret_cast<const uint16_t *>(ysm + c * (a.Cp / 2 + 1)) + an * (5 + a.nc);
    auto sg = [](float v) { return 1.0f / (1.0f + expf(-v)); };
    const float sx = sg(rbf2f(h[0])), sy = sg(rbf2f(h[1])), sw = sg(rbf2f(h[2])), sh = sg(rbf2f(h[3])), obj = sg(rbf2f(h[4]));
    const float cx = (sx * 2.f - 0.5f + (float)gx) * a.stride, cy = (sy * 2.f - 0.5f + (float)gy) * a.stride;
    const float w = (sw * 2.f) * (sw * 2.f) * a.aw[an], hh = (sh * 2.f) * (sh * 2.f) * a.ah[an];
    float best = -FLT_MAX, best_v = 0.f;
    int lab = 0;
    for (int k = 0; k < a.nc; ++k) {
        const float v = rbf2f(h[5 + k]), key = logit_key(v);
        if (key > best) { best = key; best_v = v; lab = k; }
    }
    const float conf = obj * sg(best_v);
    const size_t o = (size_t)b * a.total + a.off + (size_t)loc * a.A + an;
    *reinterpret_cast<float4 *>(boxes + o * 4) = make_float4(cx - w / 2, cy - hh / 2, cx + w / 2, cy + hh / 2);
    scores[o] = (obj > a.thr && conf > a.thr) ? conf : -FLT_MAX;
    labels[o] = lab;
}

// YOLOv8 Detect decode (anchor-free, DFL; Ultralytics v8 convention; absent from the reference: parity unpinned).
// head [B,H,W,Cp] bf16: channels [0, 4*R) = box distribution logits (side-major: l, t, r, b x R bins), [4R, 4R+nc) = class
// logits.  distance = sum_i i * softmax(bins)_i ; box = (ax - l, ay - t, ax + r, ay + b) * stride with the anchor point
// (gx + 0.5, gy + 0.5); score = max_c sigmoid(cls_c), label = first arg-max; score <= conf -> -FLT_MAX.
struct Yolo8Args { int H, W, Cp, nc, R; float stride, thr; int off, total; };
__global__ __launch_bounds__(256) void yolov8_decode_kernel(const uint16_t *__restrict__ head, Yolo8Args a, int B, int cells,
                                                            float *__restrict__ boxes, float *__restrict__ scores, int *__restrict__ labels) {
    extern __shared__ unsigned ysm[];
    const int per = a.H * a.W;
    const long long cells_total = (long long)B * per, cell0 = (long long)blockIdx.x * cells;
    stage_head_rows(head, ysm, cell0, cells_total, cells, a.Cp);   // see yolo_decode_kernel
    const int t = threadIdx.x;
    const long long cell = cell0 + t;
    if (t >= cells || cell >= cells_total) return;
    const int b = (int)(cell / per), loc = (int)(cell - (long long)b * per);
    const int gx = loc % a.W, gy = loc / a.W;
    const uint16_t *h = reinterpret_cast<const uint16_t *>(ysm + t * (a.Cp / 2 + 1));
    float d[4];
#pragma unroll
    for (int sd = 0; sd < 4; ++sd) {
        float mx = -FLT_MAX;
        for (int i = 0; i < a.R; ++i) mx = fmaxf(mx, rbf2f(h[sd * a.R + i]));
        float den = 0.f, num = 0.f;
        for (int i = 0; i < a.R; ++i) {
            const float p = expf(rbf2f(h[sd * a.R + i]) - mx);
            den += p;
            num += p * (float)i;
        }
        d[sd] = num / den;
    }
    const float ax = (float)gx + 0.5f, ay = (float)gy + 0.5f;
    float best = -FLT_MAX;
    int lab = 0;
    for (int c = 0; c < a.nc; ++c) {
        const float v = rbf2f(h[4 * a.R + c]);
        if (v > best) { best = v; lab = c; }
    }
    const float conf = 1.0f / (1.0f + expf(-best));  // sigmoid is monotonic: arg-max on the logits
    const size_t o = (size_t)b * a.total + a.off + loc;
    *reinterpret_cast<float4 *>(boxes + o * 4) =
        make_float4((ax - d[0]) * a.stride, (ay - d[1]) * a.stride, (ax + d[2]) * a.stride, (ay + d[3]) * a.stride);
    scores[o] = conf > a.thr ? conf : -FLT_MAX;
    labels[o] = lab;
}

// rotated BEV box (x, y, dx, dy, r) -> axis-aligned "standup" box of its 4 corners:
// pointpillars/src/core/box_np_ops.py:316-341 (center_to_corner_box2d, origin 0.5, corners @ [[c,-s],[s,c]])
// + :172-177 (corner_to_standup_nd); call site pointpillars/src/predict.py:61-78.
__global__ void standup_kernel(const float *__restrict__ boxes, int n, int stride, int ix, int iy, int idx_, int idy, int ir,
                               float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float *b = boxes + (size_t)i * stride;
    const float cx = b[ix], cy = b[iy], dx = b[idx_], dy = b[idy], r = b[ir];
    const float s = sinf(r), c = cosf(r);
    const float nx[4] = {-0.5f, -0.5f, 0.5f, 0.5f}, ny[4] = {-0.5f, 0.5f, 0.5f, -0.5f};
    float x0 = 0, x1 = 0, y0 = 0, y1 = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float px = dx * nx[k], py = dy * ny[k];
        const float qx = px * c + py * s + cx, qy = -px * s + py * c + cy;
        if (k == 0) { x0 = x1 = qx; y0 = y1 = qy; }
        else { x0 = fminf(x0, qx); x1 = fmaxf(x1, qx); y0 = fminf(y0, qy); y1 = fmaxf(y1, qy); }
    }
    *reinterpret_cast<float4 *>(out + (size_t)i * 4) = make_float4(x0, y0, x1, y1);
}

// out[b, j, :] = src[b, idx[b, j], :] for j < cnt[b] (zero rows past cnt); W floats per row
__global__ void gather_rows_kernel(const float *__restrict__ src, const int *__restrict__ idx, const int *__restrict__ cnt,
                                   int B, int n_src, int k, int W, float *__restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * k * W) return;
    const int w = e % W, j = (e / W) % k, b = e / (W * k);
    float v = 0.f;
    if (!cnt || j < cnt[b]) v = src[((size_t)b * n_src + idx[(size_t)b * k + j]) * W + w];
    out[e] = v;
}

__global__ void sigmoid_clip_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n, float lo, float hi) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const float s = 1.0f / (1.0f + expf(-x[e]));
        y[e] = fminf(fmaxf(s, lo), hi);
    }
}

}  // namespace md

using namespace md;

#ifdef MD_DIAG
extern "C" int md_diag_set_topk_stamp_buffer(void *p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(md::g_topk_stamps), &p, sizeof(p)) == hipSuccess ? MD_OK : MD_ERR_HIP;
}
#endif

extern "C" int md_anchors_fpn(MD_AOT_ARGS) {
    if (nparam != 1) return MD_ERR_NPARAM;
    if (!params || !extra || !dtype_is(dtypes, 0, "float32")) return MD_ERR_ARG;
    const md_fpn_anchor_attrs *at = (const md_fpn_anchor_attrs *)extra;
    if (at->num_levels < 1 || at->num_levels > 8 || at->num_ratios < 1 || at->num_ratios > 16) return MD_ERR_ARG;
    FpnArgs a;
    a.L = at->num_levels;
    a.A = at->num_ratios;
    int64_t off = 0;
    for (int l = 0; l < a.L; ++l) {
        a.lv[l].H = at->feat_h[l]; a.lv[l].W = at->feat_w[l]; a.lv[l].stride = at->stride[l];
        a.lv[l].offset = off;
        off += (int64_t)at->feat_h[l] * at->feat_w[l] * a.A;
        for (int r = 0; r < a.A; ++r) {
            // identical float32 op order to the oracle (np_ops.fpn_anchors)
            const float hr = sqrtf(at->ratios[r]);
            const float wr = 1.0f / hr;
            const float ws = (float)at->stride[l] * wr * at->scale;
            const float hs = (float)at->stride[l] * hr * at->scale;
            a.base[l][r][0] = -0.5f * ws; a.base[l][r][1] = -0.5f * hs;
            a.base[l][r][2] = 0.5f * ws;  a.base[l][r][3] = 0.5f * hs;
        }
    }
    if (numel(ndims, shapes, 0) != off * 4) return MD_ERR_ARG;
    if (off == 0) return MD_OK;
    hipLaunchKernelGGL(anchors_fpn_kernel, dim3(grid1d((size_t)off)), dim3(256), 0, (hipStream_t)stream, a,
                       (float4 *)params[0], (size_t)off);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_anchors_3d_stride(MD_AOT_ARGS) {
    if (nparam != 1) return MD_ERR_NPARAM;
    if (!params || !extra || !dtype_is(dtypes, 0, "float32")) return MD_ERR_ARG;
    const md_anchor3d_attrs *at = (const md_anchor3d_attrs *)extra;
    if (at->feat_h < 2 || at->feat_w < 2 || at->num_rot < 1 || at->num_rot > 8) return MD_ERR_ARG;
    Anchor3dArgs a;
    a.H = at->feat_h; a.W = at->feat_w; a.R = at->num_rot;
    // np.arange(start, stop, step, dtype=float32): first = f32(start), second = f32(start + step) (double add),
    // delta = second - first in float32, v[i] = first + i*delta   (numpy FLOAT_fill)
    const double xs = ((double)at->range[3] - (double)at->range[0]) / (double)(at->feat_w - 1);
    const double ys = ((double)at->range[4] - (double)at->range[1]) / (double)(at->feat_h - 1);
    a.x_first = (float)(double)at->range[0];
    a.x_delta = (float)((double)at->range[0] + xs) - a.x_first;
    a.y_first = (float)(double)at->range[1];
    a.y_delta = (float)((double)at->range[1] + ys) - a.y_first;
    a.z = (float)at->z_offset;
    for (int i = 0; i < 3; ++i) a.size[i] = (float)at->size[i];
    for (int i = 0; i < a.R; ++i) a.rot[i] = (float)at->rotations[i];
    const int64_t total = (int64_t)a.H * a.W * a.R;
    // standalone (slots_total 0): the table of a location has R rows; inside a concat of generators (target_assigner.py:242)
    // this generator fills rows [slot_off, slot_off + R) of slots_total
    a.slots = at->slots_total > 0 ? at->slots_total : a.R;
    a.slot_off = at->slots_total > 0 ? at->slot_off : 0;
    if (a.slot_off < 0 || a.slot_off + a.R > a.slots) return MD_ERR_ARG;
    if (numel(ndims, shapes, 0) != (int64_t)a.H * a.W * a.slots * 7) return MD_ERR_ARG;
    hipLaunchKernelGGL(anchors_3d_stride_kernel, dim3(grid1d((size_t)total)), dim3(256), 0, (hipStream_t)stream, a,
                       (float *)params[0]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_anchors_3d_range(MD_AOT_ARGS) {
    if (nparam != 1) return MD_ERR_NPARAM;
    if (!params || !extra || !dtype_is(dtypes, 0, "float32")) return MD_ERR_ARG;
    const md_anchor3d_range_attrs *at = (const md_anchor3d_range_attrs *)extra;
    if (at->feat_d < 1 || at->feat_h < 1 || at->feat_w < 1 || at->num_sizes < 1 || at->num_sizes > 4 || at->num_rot < 1 ||
        at->num_rot > 8 || (at->linspace_mode != 0 && at->linspace_mode != 1))
        return MD_ERR_ARG;
    Anchor3dRangeArgs a;
    a.D = at->feat_d; a.H = at->feat_h; a.W = at->feat_w; a.S = at->num_sizes; a.R = at->num_rot; a.mode = at->linspace_mode;
    const int n_ax[3] = {a.W, a.H, a.D};
    for (int i = 0; i < 3; ++i) {   // anchor_range = np.array(anchor_range, float32); linspace(range[i], range[i + 3], n)
        LinAxis &ax = a.ax[i];
        ax.lo = (float)at->range[i]; ax.hi = (float)at->range[i + 3]; ax.n = n_ax[i];
        const float delta = ax.hi - ax.lo;
        ax.step = ax.n > 1 ? delta / (float)(ax.n - 1) : 0.f;
    }
    for (int k = 0; k < a.S; ++k)
        for (int i = 0; i < 3; ++i) a.size[k][i] = (float)at->sizes[k][i];
    for (int i = 0; i < a.R; ++i) a.rot[i] = (float)at->rotations[i];
    const int per = a.S * a.R;
    a.slots = at->slots_total > 0 ? at->slots_total : per;
    a.slot_off = at->slots_total > 0 ? at->slot_off : 0;
    if (a.slot_off < 0 || a.slot_off + per > a.slots) return MD_ERR_ARG;
    const int64_t locs = (int64_t)a.D * a.H * a.W;
    if (numel(ndims, shapes, 0) != locs * a.slots * 7) return MD_ERR_ARG;
    if (!params[0]) return MD_ERR_ARG;
    hipLaunchKernelGGL(anchors_3d_range_kernel, dim3(grid1d((size_t)locs * per)), dim3(256), 0, (hipStream_t)stream, a, (float *)params[0]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_anchor_mask(MD_AOT_ARGS) {
    // in: coors[V,3] i32 (z,y,x), anchors_bv[N,4] f32 ; out: area[N] f32, mask[N] u8 ; ws: ny*nx i32
    if (nparam != 4 && nparam != 5) return MD_ERR_NPARAM;
    if (!params || !extra) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "int32") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "float32") ||
        !dtype_is(dtypes, 3, "uint8"))
        return MD_ERR_ARG;
    const md_anchor_mask_attrs *at = (const md_anchor_mask_attrs *)extra;
    const int64_t nv = dim(ndims, shapes, 0, 0), n = dim(ndims, shapes, 1, 0);
    if (nv < 0 || n < 0 || dim(ndims, shapes, 0, 1) != 3 || dim(ndims, shapes, 1, 1) != 4) return MD_ERR_ARG;
    const int nx = at->grid_x, ny = at->grid_y;
    if (nx < 1 || ny < 1 || (int64_t)nx * ny > (1 << 28)) return MD_ERR_SIZE;
    hipStream_t s = (hipStream_t)stream;
    Scratch ws;
    int rc = ws.acquire((size_t)nx * ny * 4, nparam, params, ndims, shapes, 4, s);
    if (rc) return rc;
    int *dense = (int *)ws.ptr;
    MD_HIP_TRY(hipMemsetAsync(dense, 0, (size_t)nx * ny * 4, s));
    if (nv > 0)
        hipLaunchKernelGGL(amask_scatter_kernel, dim3((unsigned)((nv + 255) / 256)), dim3(256), 0, s,
                           (const int *)params[0], (int)nv, nx, ny, dense);
    hipLaunchKernelGGL(amask_cumsum_y_kernel, dim3((nx + 255) / 256), dim3(256), 0, s, dense, nx, ny);
    hipLaunchKernelGGL(amask_cumsum_x_kernel, dim3((ny + 3) / 4), dim3(256), 0, s, dense, nx, ny);
    if (n > 0)
        hipLaunchKernelGGL(amask_area_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dense,
                           (const float *)params[1], (int)n, nx, ny, at->voxel_x, at->voxel_y, at->offset_x,
                           at->offset_y, at->area_threshold, (float *)params[2], (unsigned char *)params[3]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_second_box_decode(MD_AOT_ARGS) {
    // in: encodings[..., 7] f32, anchors[A,7] f32 (broadcast over leading dims: row i uses anchor i % A) ; out: boxes
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params) return MD_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (!dtype_is(dtypes, i, "float32")) return MD_ERR_ARG;
    const int64_t tot = numel(ndims, shapes, 0), ta = numel(ndims, shapes, 1);
    if (tot < 0 || ta <= 0 || tot % 7 || ta % 7 || numel(ndims, shapes, 2) != tot || (tot / 7) % (ta / 7)) return MD_ERR_ARG;
    if (tot == 0) return MD_OK;
    hipLaunchKernelGGL(second_box_decode_kernel, dim3(grid1d((size_t)tot / 7)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (const float *)params[1], (size_t)tot / 7, (size_t)ta / 7,
                       (float *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_delta2bbox(MD_AOT_ARGS) {
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !extra) return MD_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (!dtype_is(dtypes, i, "float32")) return MD_ERR_ARG;
    const int64_t tot = numel(ndims, shapes, 0);
    if (tot < 0 || tot % 4 || numel(ndims, shapes, 1) != tot || numel(ndims, shapes, 2) != tot) return MD_ERR_ARG;
    const md_delta2bbox_attrs *at = (const md_delta2bbox_attrs *)extra;
    DeltaArgs a;
    for (int i = 0; i < 4; ++i) { a.mean[i] = at->means[i]; a.stdv[i] = at->stds[i]; }
    a.max_ratio = at->max_ratio; a.clip_w = at->clip_w; a.clip_h = at->clip_h; a.do_clip = at->clip_w > 0 && at->clip_h > 0;
    if (tot == 0) return MD_OK;
    hipLaunchKernelGGL(delta2bbox_kernel, dim3(grid1d((size_t)tot / 4)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (const float *)params[1], (size_t)tot / 4, a, (float *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_topk_segmented(MD_AOT_ARGS) {
    // in: scores[T] f32, seg_off[L+1] i32 ; out: values[L,k] f32, indices[L,k] i32, count[L] i32
    if (nparam != 5 && nparam != 6) return MD_ERR_NPARAM;
    if (!params || !extra) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "int32") || !dtype_is(dtypes, 2, "float32") ||
        !dtype_is(dtypes, 3, "int32") || !dtype_is(dtypes, 4, "int32"))
        return MD_ERR_ARG;
    const md_topk_attrs *at = (const md_topk_attrs *)extra;
    const int64_t L = numel(ndims, shapes, 1) - 1;
    if (L < 0 || at->k < 1) return MD_ERR_ARG;
    if (at->k > TOPK_MAXK) return MD_ERR_SIZE;
    if (numel(ndims, shapes, 2) != L * at->k || numel(ndims, shapes, 3) != L * at->k || numel(ndims, shapes, 4) != L)
        return MD_ERR_ARG;
    if (L == 0) return MD_OK;
    hipStream_t s = (hipStream_t)stream;
    // (r03: lowering the threshold to one 8192-score chunk -- the 32-image YOLO batches, 25 200 / 8 400 scores per image, run the single-workgroup
    // form for 110 / 86 us -- measured no gain: three launches + a memset cost what the idle CUs cost)
    if (at->max_segment > 4 * TK_CHUNK && at->k <= TK_CAP / 2 && L <= 65535) {
        const size_t hist_bytes = align_up((size_t)L * TK_BINS * 4 + (size_t)L * 4, 256);
        Scratch ws;
        int rc = ws.acquire(hist_bytes + (size_t)L * TK_CAP * 8, nparam, params, ndims, shapes, 5, s);
        if (rc) return rc;
        unsigned *ghist = (unsigned *)ws.ptr, *cand_cnt = ghist + (size_t)L * TK_BINS;
        unsigned long long *cand = (unsigned long long *)((char *)ws.ptr + hist_bytes);
        MD_HIP_TRY(hipMemsetAsync(ws.ptr, 0, hist_bytes, s));
        const unsigned chunks = (unsigned)((at->max_segment + TK_CHUNK - 1) / TK_CHUNK);
        hipLaunchKernelGGL(topk_hist_kernel, dim3(chunks, (unsigned)L), dim3(256), 0, s, (const float *)params[0],
                           (const int *)params[1], at->min_score, ghist);
        hipLaunchKernelGGL(topk_compact_kernel, dim3(chunks, (unsigned)L), dim3(256), 0, s, (const float *)params[0],
                           (const int *)params[1], at->k, at->min_score, ghist, cand_cnt, cand);
        hipLaunchKernelGGL(topk_final_kernel, dim3((unsigned)L), dim3(TOPK_THREADS), 0, s, (const float *)params[0],
                           (const int *)params[1], at->k, at->min_score, cand_cnt, cand, (float *)params[2],
                           (int *)params[3], (int *)params[4]);
    } else if (at->max_segment > 0 && at->max_segment <= TOPK_LDS_MAX_N) {
        const int cap = at->max_segment;
        const int lds = (int)align_up((size_t)cap * 4, 16);
        if (ensure_dyn_lds((const void *)topk_segmented_lds_kernel, lds) != MD_OK) return MD_ERR_HIP;
        hipLaunchKernelGGL(topk_segmented_lds_kernel, dim3((unsigned)L), dim3(TOPK_THREADS), lds, s, (const float *)params[0],
                           (const int *)params[1], at->k, at->min_score, (float *)params[2], (int *)params[3], (int *)params[4], cap);
    } else {
        hipLaunchKernelGGL(topk_segmented_kernel, dim3((unsigned)L), dim3(TOPK_THREADS), 0, s, (const float *)params[0],
                           (const int *)params[1], at->k, at->min_score, (float *)params[2], (int *)params[3],
                           (int *)params[4]);
    }
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_roi_align(MD_AOT_ARGS) {
    // in: rois[R,5] f32, feat_0..feat_{L-1} [N,H,W,C] bf16 ; out: pooled[R,P,P,C] bf16, level[R] i32 (may be NULL ptr)
    if (!params || !extra) return MD_ERR_ARG;
    const md_roi_align_attrs *at = (const md_roi_align_attrs *)extra;
    const int L = at->num_levels;
    if (L < 1 || L > 6) return MD_ERR_ARG;
    if (nparam != 1 + L + 2) return MD_ERR_NPARAM;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1 + L, "bfloat16") || !dtype_is(dtypes, 2 + L, "int32"))
        return MD_ERR_ARG;
    const int64_t R = dim(ndims, shapes, 0, 0);
    if (R < 0 || dim(ndims, shapes, 0, 1) != 5) return MD_ERR_ARG;
    RoiArgs a;
    a.L = L; a.P = at->pooled; a.sampling = at->sampling_ratio; a.aligned = at->aligned;
    a.k_min = at->k_min; a.canonical_level = at->canonical_level; a.canonical_scale = at->canonical_scale;
    if (a.P < 1 || a.sampling < 1) return MD_ERR_ARG;
    for (int j = 0; j < 5; ++j) {
        const double edge = (double)at->canonical_scale * (ldexp(1.0, at->k_min + j + 1 - at->canonical_level) - 1e-6);
        a.lvl_thr[j] = (float)(edge * edge);
    }
    a.C = 0; a.N = 0;
    for (int l = 0; l < L; ++l) {
        if (!dtype_is(dtypes, 1 + l, "bfloat16") || ndims[1 + l] != 4) return MD_ERR_ARG;
        a.lv[l].feat = (const uint16_t *)params[1 + l];
        a.lv[l].H = (int)shapes[1 + l][1]; a.lv[l].W = (int)shapes[1 + l][2];
        a.lv[l].scale = at->spatial_scale[l];
        if (l == 0) { a.C = (int)shapes[1][3]; a.N = (int)shapes[1][0]; }
        else if (shapes[1 + l][3] != a.C || shapes[1 + l][0] != a.N) return MD_ERR_ARG;
    }
    if (a.C % 8) return MD_ERR_ARG;
    if (numel(ndims, shapes, 1 + L) != R * a.P * a.P * a.C) return MD_ERR_ARG;
    if (R == 0) return MD_OK;
    const size_t total = (size_t)R * a.P * a.P * (a.C / 8);
    if (a.C % 32 == 0)
        hipLaunchKernelGGL(roi_align_c32_kernel, dim3(grid1d(total / 4)), dim3(256), 0, (hipStream_t)stream, a,
                           (const float *)params[0], (int)R, (uint16_t *)params[1 + L], (int *)params[2 + L]);
    else
        hipLaunchKernelGGL(roi_align_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream, a,
                           (const float *)params[0], (int)R, (uint16_t *)params[1 + L], (int *)params[2 + L]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_heat_nms(MD_AOT_ARGS) {
    // in: heat[B,C,H,W] f32 ; out: same shape
    if (nparam != 2) return MD_ERR_NPARAM;
    if (!params || !ndims || ndims[0] != 4 || !dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32")) return MD_ERR_ARG;
    if (numel(ndims, shapes, 1) != numel(ndims, shapes, 0)) return MD_ERR_ARG;
    const size_t planes = (size_t)shapes[0][0] * shapes[0][1];
    const int H = (int)shapes[0][2], W = (int)shapes[0][3];
    if (planes * H * W == 0) return MD_OK;
    hipLaunchKernelGGL(heat_nms_kernel, dim3(grid1d(planes * H * W)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (float *)params[1], H, W, planes);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_heat_peaks(MD_AOT_ARGS) {
    // in: head[B,H,W,Cp] bf16 ; out: heat[B,nc,H,W] f32 (sigmoid + clip, zeroed where not a 3x3 maximum), hm[B,nc,H,W] f32 | NULL
    if (nparam != 3) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4 || !dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32") ||
        !dtype_is(dtypes, 2, "float32"))
        return MD_ERR_ARG;
    const md_heat_peaks_attrs *at = (const md_heat_peaks_attrs *)extra;
    PeakArgs a;
    const int64_t B = shapes[0][0];
    a.H = (int)shapes[0][1]; a.W = (int)shapes[0][2]; a.Cp = (int)shapes[0][3];
    a.c0 = at->c0; a.nc = at->num_classes; a.lo = at->lo; a.hi = at->hi;
    if (a.Cp % 8 || a.c0 < 0 || a.c0 % 8 || a.nc < 1 || a.c0 + (a.nc + 7) / 8 * 8 > a.Cp) return MD_ERR_ARG;
    if (numel(ndims, shapes, 1) != B * a.nc * a.H * a.W || (params[2] && numel(ndims, shapes, 2) != B * a.nc * a.H * a.W)) return MD_ERR_ARG;
    if (B * a.H * a.W == 0) return MD_OK;
    if (!params[0] || !params[1]) return MD_ERR_ARG;
    const long long tiles = (long long)((a.H + PK_TH - 1) / PK_TH) * ((a.W + PK_TW - 1) / PK_TW);
    if (tiles > 0x7fffffffLL || B > 65535 || (a.nc + PK_C - 1) / PK_C > 65535) return MD_ERR_SIZE;
    hipLaunchKernelGGL(heat_peaks_kernel, dim3((unsigned)tiles, (a.nc + PK_C - 1) / PK_C, (unsigned)B), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], a, (float *)params[1], (float *)params[2]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_centernet_assemble(MD_AOT_ARGS) {
    // in: top_score[B,K] f32, top_ind2[B,K] i32, cls_inds[B,C,K] i32, wh[B,2,H,W] f32, reg[B,2,H,W] f32 | NULL
    // out: det[B,K,6] f32, inds[B,K] i32, cls[B,K] i32
    if (nparam != 8) return MD_ERR_NPARAM;
    if (!params || !ndims || ndims[0] != 2 || ndims[2] != 3 || ndims[3] != 4) return MD_ERR_ARG;
    const int B = (int)shapes[0][0], K = (int)shapes[0][1], C = (int)shapes[2][1];
    const int H = (int)shapes[3][2], W = (int)shapes[3][3];
    if (shapes[2][2] != K || shapes[2][0] != B || shapes[3][0] != B || shapes[3][1] != 2) return MD_ERR_ARG;
    if (numel(ndims, shapes, 5) != (int64_t)B * K * 6) return MD_ERR_ARG;
    if (B * K == 0) return MD_OK;
    hipLaunchKernelGGL(centernet_assemble_kernel, dim3((B * K + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (const int *)params[1], (const int *)params[2], (const float *)params[3],
                       (const float *)params[4], B, C, K, H, W, (float *)params[5], (int *)params[6], (int *)params[7]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_centerpoint_decode(MD_AOT_ARGS) {
    // in: head[B,H,W,C] bf16 ; out: scores[B,HW] f32, labels[B,HW] i32, boxes[B,HW,9] f32, nms_boxes[B,HW,7] f32
    if (nparam != 5) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || ndims[0] != 4) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "float32") || !dtype_is(dtypes, 4, "float32"))
        return MD_ERR_ARG;
    const md_centerpoint_attrs *at = (const md_centerpoint_attrs *)extra;
    CpArgs a;
    a.o_reg = at->off_reg; a.o_height = at->off_height; a.o_dim = at->off_dim; a.o_rot = at->off_rot;
    a.o_vel = at->off_vel; a.o_hm = at->off_hm; a.ncls = at->num_classes;
    const int B = (int)shapes[0][0];
    a.H = (int)shapes[0][1]; a.W = (int)shapes[0][2]; a.C = (int)shapes[0][3];
    if (a.ncls < 1 || a.o_hm + a.ncls > a.C || a.o_reg + 2 > a.C || a.o_dim + 3 > a.C || a.o_rot + 2 > a.C ||
        a.o_height + 1 > a.C || a.o_vel + 2 > a.C)
        return MD_ERR_ARG;
    a.score_thr = at->score_threshold; a.osf = at->out_size_factor; a.vx = at->voxel_size[0]; a.vy = at->voxel_size[1];
    a.px = at->pc_range[0]; a.py = at->pc_range[1];
    for (int i = 0; i < 3; ++i) { a.rmin[i] = at->post_center_range[i]; a.rmax[i] = at->post_center_range[3 + i]; }
    const int64_t total = (int64_t)B * a.H * a.W;
    if (numel(ndims, shapes, 1) != total || numel(ndims, shapes, 2) != total || numel(ndims, shapes, 3) != total * 9 ||
        numel(ndims, shapes, 4) != total * 7)
        return MD_ERR_ARG;
    if (total == 0) return MD_OK;
    if (total > 0x7fffffffLL) return MD_ERR_SIZE;
    hipLaunchKernelGGL(centerpoint_decode_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint16_t *)params[0], a, (int)total, (float *)params[1], (int *)params[2], (float *)params[3],
                       (float *)params[4]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_yolo_decode(MD_AOT_ARGS) {
    if (nparam != 4) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || ndims[0] != 4) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "float32") ||
        !dtype_is(dtypes, 3, "int32"))
        return MD_ERR_ARG;
    const md_yolo_attrs *at = (const md_yolo_attrs *)extra;
    YoloArgs a;
    const int B = (int)shapes[0][0];
    a.H = (int)shapes[0][1]; a.W = (int)shapes[0][2]; a.Cp = (int)shapes[0][3];
    a.nc = at->num_classes; a.A = at->num_anchors; a.stride = at->stride; a.thr = at->conf_thres;
    a.off = at->out_offset; a.total = at->out_total;
    if (a.A < 1 || a.A > 3 || a.nc < 1 || a.A * (5 + a.nc) > a.Cp) return MD_ERR_ARG;
    for (int i = 0; i < a.A; ++i) { a.aw[i] = at->anchors[2 * i]; a.ah[i] = at->anchors[2 * i + 1]; }
    const int64_t per = (int64_t)a.H * a.W * a.A;
    if (a.off < 0 || a.off + per > a.total) return MD_ERR_ARG;
    if (numel(ndims, shapes, 1) != (int64_t)B * a.total * 4 || numel(ndims, shapes, 2) != (int64_t)B * a.total ||
        numel(ndims, shapes, 3) != (int64_t)B * a.total)
        return MD_ERR_ARG;
    if (B * per == 0) return MD_OK;
    if (a.Cp % 8) return MD_ERR_ARG;
    // cells per 256-thread workgroup: A threads per cell, rows of Cp/2 + 1 dwords in at most 60 KiB of LDS
    int cells = 256 / a.A;
    if (cells > 64) cells = 64;
    const int rowb = (a.Cp / 2 + 1) * 4;
    if (cells * rowb > 60 * 1024) cells = 60 * 1024 / rowb;
    if (cells < 1) return MD_ERR_SIZE;
    const long long cells_total = (long long)B * a.H * a.W;
    hipLaunchKernelGGL(yolo_decode_kernel, dim3((unsigned)((cells_total + cells - 1) / cells)), dim3(256), (size_t)cells * rowb,
                       (hipStream_t)stream, (const uint16_t *)params[0], a, B, cells, (float *)params[1], (float *)params[2],
                       (int *)params[3]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_standup_boxes(MD_AOT_ARGS) {
    // in: boxes[N, S] f32 (S = 5: x,y,dx,dy,r ; S = 7: x,y,z,dx,dy,dz,r) ; out: standup[N,4] f32
    if (nparam != 2) return MD_ERR_NPARAM;
    if (!params || !ndims || ndims[0] != 2 || !dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32")) return MD_ERR_ARG;
    const int64_t n = shapes[0][0];
    const int S = (int)shapes[0][1];
    if ((S != 5 && S != 7) || numel(ndims, shapes, 1) != n * 4) return MD_ERR_ARG;
    if (n == 0) return MD_OK;
    if (n > 0x7fffffffLL) return MD_ERR_SIZE;
    if (S == 5)
        hipLaunchKernelGGL(standup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)params[0], (int)n, 5, 0, 1, 2, 3, 4, (float *)params[1]);
    else
        hipLaunchKernelGGL(standup_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float *)params[0], (int)n, 7, 0, 1, 3, 4, 6, (float *)params[1]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_gather_rows(MD_AOT_ARGS) {
    // in: src[B,n,W] f32, idx[B,k] i32, cnt[B] i32 or NULL ; out: out[B,k,W] f32
    if (nparam != 4) return MD_ERR_NPARAM;
    if (!params || !ndims || ndims[0] != 3 || ndims[1] != 2) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "int32") || !dtype_is(dtypes, 2, "int32") ||
        !dtype_is(dtypes, 3, "float32"))
        return MD_ERR_ARG;
    const int B = (int)shapes[0][0], n = (int)shapes[0][1], W = (int)shapes[0][2], k = (int)shapes[1][1];
    if (shapes[1][0] != B || numel(ndims, shapes, 3) != (int64_t)B * k * W) return MD_ERR_ARG;
    if ((int64_t)B * k * W == 0) return MD_OK;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)(((int64_t)B * k * W + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, (const float *)params[0], (const int *)params[1], (const int *)params[2], B, n, k,
                       W, (float *)params[3]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_sigmoid_clip(MD_AOT_ARGS) {
    if (nparam != 2) return MD_ERR_NPARAM;
    if (!params || !dtype_is(dtypes, 0, "float32") || !dtype_is(dtypes, 1, "float32")) return MD_ERR_ARG;
    const int64_t n = numel(ndims, shapes, 0);
    if (n < 0 || numel(ndims, shapes, 1) != n) return MD_ERR_ARG;
    float lo = 1e-4f, hi = 1.0f - 1e-4f;
    if (extra) { lo = ((const md_clip_attrs *)extra)->lo; hi = ((const md_clip_attrs *)extra)->hi; }
    if (n == 0) return MD_OK;
    hipLaunchKernelGGL(sigmoid_clip_kernel, dim3(grid1d((size_t)n)), dim3(256), 0, (hipStream_t)stream,
                       (const float *)params[0], (float *)params[1], (size_t)n, lo, hi);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

extern "C" int md_yolov8_decode(MD_AOT_ARGS) {
    // in: head[B,H,W,Cp >= 4*reg_max + nc] bf16 ; out: boxes[B,total,4] f32, scores[B,total] f32, labels[B,total] i32
    if (nparam != 4) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || ndims[0] != 4) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "float32") || !dtype_is(dtypes, 2, "float32") || !dtype_is(dtypes, 3, "int32"))
        return MD_ERR_ARG;
    const md_yolov8_attrs *at = (const md_yolov8_attrs *)extra;
    Yolo8Args a;
    const int B = (int)shapes[0][0];
    a.H = (int)shapes[0][1]; a.W = (int)shapes[0][2]; a.Cp = (int)shapes[0][3];
    a.nc = at->num_classes; a.R = at->reg_max; a.stride = at->stride; a.thr = at->conf_thres; a.off = at->out_offset; a.total = at->out_total;
    const int per = a.H * a.W;
    if (a.nc < 1 || a.R < 1 || a.R > 64 || 4 * a.R + a.nc > a.Cp || a.off < 0 || a.off + per > a.total) return MD_ERR_ARG;
    if (numel(ndims, shapes, 1) != (int64_t)B * a.total * 4 || numel(ndims, shapes, 2) != (int64_t)B * a.total ||
        numel(ndims, shapes, 3) != (int64_t)B * a.total)
        return MD_ERR_ARG;
    if ((int64_t)B * per == 0) return MD_OK;
    if ((int64_t)B * per > 0x7fffffffLL) return MD_ERR_SIZE;
    if (a.Cp % 8) return MD_ERR_ARG;
    int cells = 128;                                  // one thread per cell; rows of Cp/2 + 1 dwords in at most 60 KiB of LDS
    const int rowb = (a.Cp / 2 + 1) * 4;
    if (cells * rowb > 60 * 1024) cells = 60 * 1024 / rowb;
    if (cells < 1) return MD_ERR_SIZE;
    const long long cells_total = (long long)B * per;
    hipLaunchKernelGGL(yolov8_decode_kernel, dim3((unsigned)((cells_total + cells - 1) / cells)), dim3(256), (size_t)cells * rowb,
                       (hipStream_t)stream, (const uint16_t *)params[0], a, B, cells, (float *)params[1], (float *)params[2],
                       (int *)params[3]);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}

