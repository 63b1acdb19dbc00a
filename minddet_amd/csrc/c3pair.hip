// c3pair.hip -- the Bottleneck of a YOLOv5 C3 block (Conv 1x1 + BN + SiLU -> Conv 3x3 + BN + SiLU, + shortcut) in ONE launch.
//
// What it replaces: the two md_conv2d launches per (cv1, cv2) pair of graphs.C3 (build-authored YOLOv5 of BASELINE configs[1]; the
// reference names the model family only, /root/reference/README.md:5-14).  At the benchmark's shard (32 images, 640 x 640) these layers
// are 18-29 us launches whose tensors never leave the Infinity Cache: what they pay for is the launch itself -- ramp-up, one exposed
// L2 round trip per tile and the tail -- and the second launch pays it again (profiles/r04_yolov5s_conv_layers.json: 0.17-0.30 of the
// layer-wise roofline).  Here a workgroup owns an 8 x 16 block of output pixels of one image:
//
//   phase A  T1[10 x 18 halo px][C] = silu(W1 . x + b1), zero outside the image (the 3x3 conv's zero padding applies to T1); the x halo
//            tile is DMAed once (LDS-DMA), T1 overwrites it in place
//   phase B  acc[8 x 16 px][C] = sum over (64-channel chunk, tap) of W2[chunk, tap] . T1[rows shifted by the tap]
//   epilogue y = bf16(silu(acc + b2)) (+ x, the shortcut, added in fp32 and rounded once more -- exactly what md_conv2d's residual path
//            does), written as 16-B pieces; optionally the NEXT C channels of x are copied to the next C channels of y (the cv2(x) half of
//            the C3 concat buffer, when the block's output has to land in another buffer than its input: a fused pair cannot run in place,
//            its halo pixels are another workgroup's outputs)
//
// The weights do not fit LDS (C = 128: W2 is 288 KiB): W1 and W2 travel through a ring of [C couts][64 k] sub-units (16 KiB at C = 128; two
// slots there, three at C = 64), the next one(s) in flight while one is multiplied, one raw barrier per sub-unit.  The ring is kept this
// small on purpose: what a tile costs is its chain of exposed latencies (x tile, weights, shortcut values, stores), and only a second /
// third resident workgroup hides them (r04, tools/c3pair_check.py: a deeper ring with ONE workgroup per CU was no faster than two launches).  K order = md_conv2d's for these layers (1x1: k;
// 3x3 with korder 1: (ci / 64, tap, ci % 64)), same MFMA shape, same bf16 rounding points: bit-identical to the two launches.
//
// LDS: T (192 rows x 2 C B; b1 | b2 in its unused last rows) | ring = 48 KiB (C = 64: three workgroups per CU) / 80 KiB (C = 128: two).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

namespace md {

typedef __attribute__((ext_vector_type(8))) short cp_bf16x8;
typedef __attribute__((ext_vector_type(16))) float cp_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int cp_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int cp_u32x2;
typedef float cp_f32x2 __attribute__((ext_vector_type(2)));
typedef float cp_f32x4 __attribute__((ext_vector_type(4)));   // (LDS accesses use builtin vector types only: see bottleneck.hip)

struct C3PairArgs {
    const uint16_t *x;    // [N,H,W,XC]; the pair reads channels [x_off, x_off + C) (+ the next C when `pass`)
    const uint16_t *w1;   // [C][C]
    const uint16_t *w2;   // [C][9 C], K = (ci / 64) * 576 + tap * 64 + ci % 64
    const float *b12;     // b1 | b2 (C floats each)
    uint16_t *y;          // [N,H,W,YC]; written: channels [y_off, y_off + C) (+ the next C when `pass`)
    int N, H, W, XC, YC, x_off, y_off, shortcut, pass;
    int tiles_x, tiles_y, n_tiles, pt_per_xcd;
    unsigned x_bytes, y_bytes;
    int w1_ld, w2_ld;          // elements per packed weight row (C = 32: 64 / 320, md_conv2d's K padding)
    unsigned long long *dbg;   // MD_DIAG builds only
};

constexpr int CP_TH = 8, CP_TW = 16, CP_HW = CP_TW + 2, CP_HALO = (CP_TH + 2) * CP_HW, CP_ROWS = 192;

__device__ __forceinline__ unsigned cp_pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// the SiLU of md_conv2d's epilogues (conv.hip: hardware reciprocal)
__device__ __forceinline__ float cp_silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// MD_DIAG build (tools/c3pair_stamps.py): cycle stamps of one mid-grid workgroup
#ifdef MD_DIAG
#define CP_STAMP(I) do { if (a.dbg) { __builtin_amdgcn_sched_barrier(0); stp[I] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define CP_STAMP(I) do { } while (0)
#endif

template <int C>
struct CpGeom {
    static constexpr int ROWB = 2 * C;              // bytes of a T row
    static constexpr int NCH = C / 8;               // 16-B chunks per T row
    static constexpr int NWC = C / 32;              // cout fragments = waves along cout
    static constexpr int NWQ = 8 / NWC;             // wave groups along pixels
    static constexpr int NFA = (6 + NWQ - 1) / NWQ; // halo-row fragments per wave in phase A (6 in all)
    static constexpr int NFB = 4 / NWQ;             // pixel fragments per wave in phase B (4 in all)
    static constexpr int NW1 = C / 64;              // sub-units ([C couts][64 k]) of W1
    static constexpr int NU = NW1 * 9;              // sub-units of W2: (64-channel chunk, tap)
    static constexpr int S = NW1 + NU;              // the weight stream: W1's sub-units, then W2's, in K order
    static constexpr int UNIT = C * 128;            // bytes of a sub-unit
    static constexpr int PPW = C / 64;              // DMA pieces (8 rows x 128 B) per wave and sub-unit
    static constexpr int XPW = CP_ROWS * ROWB / 1024 / 8;   // DMA pieces of the x halo tile per wave
    // the ring moves groups of GS consecutive sub-units through NSLOT slots, NSLOT - 1 groups in flight ahead of the one being multiplied
    static constexpr int GS = 1, NSLOT = C == 64 ? 3 : 2;
    static constexpr int NG = (S + GS - 1) / GS;
    static constexpr int g_first(int g) { return g * GS; }
    static constexpr int g_count(int g) { return S - g * GS < GS ? S - g * GS : GS; }
    // this wave's DMA instructions of groups g + 1 .. g + NSLOT - 2: what may stay in flight when group g is needed
    static constexpr int younger(int g) {
        int n = 0;
        for (int k = g + 1; k <= g + NSLOT - 2 && k < NG; ++k) n += g_count(k) * PPW;
        return n;
    }
    // b1 | b2 live in T's rows 184..191: the halo has 180 rows, the last fragment's rows beyond it are neither staged nor written
    static constexpr int BIAS_ROW = 184;
    static constexpr int T_OFF = 0, BIAS = BIAS_ROW * ROWB, RING = CP_ROWS * ROWB, LDS = RING + NSLOT * GS * UNIT;
    static constexpr int ES = ROWB + 16;            // row stride of the epilogue's [pixel][cout] image (over T)
    static constexpr int EP_ITERS = 128 * NCH / 512;
    static_assert(8 * C <= (CP_ROWS - BIAS_ROW) * ROWB && 128 * ES <= BIAS, "LDS map");
};

// T rows are XOR-swizzled in 16-B chunks so that a ds_read_b128 lane group (16 lanes, 16 different rows, one logical chunk) covers all
// 64 banks once: 128-B rows pair up (key (row >> 1) & 7, as in conv.hip / bottleneck.hip), 256-B rows take the row's own low bits
template <int C>
__device__ __forceinline__ int cp_key(int row) { return C == 64 ? ((row >> 1) & 7) : (row & 15); }
template <int C>
__device__ __forceinline__ int cp_swz(int row, int chunk) { return row * (2 * C) + ((chunk ^ cp_key<C>(row)) << 4); }
__device__ __forceinline__ int cp_wswz(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }   // sub-units: 128-B rows

template <int C>
__device__ __forceinline__ void c3pair_body(const C3PairArgs &a) {
    typedef CpGeom<C> G;
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr unsigned OOR = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    // lane -> pixel of a 32-pixel fragment (two 16-pixel tile rows) where the pixel operand is read at tap-shifted halo rows: every
    // ds_read_b128 lane group {0-3, 12-15, 20-27} / {4-11, 16-19, 28-31} gets one whole tile row = 16 consecutive halo rows (bottleneck.hip)
    const int hp = (int)(((0x73261540u >> ((lr >> 2) * 4)) & 7u) << 2) | (lr & 3);
    const int wc = wave % G::NWC, wq = wave / G::NWC;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pt = xcd * a.pt_per_xcd + slot;
    if (pt >= a.n_tiles) return;
    const int tx = pt % a.tiles_x, ty = (pt / a.tiles_x) % a.tiles_y, n = pt / (a.tiles_x * a.tiles_y);
    const int y0 = ty * CP_TH, x0 = tx * CP_TW;
#ifdef MD_DIAG
    unsigned long long stp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    CP_STAMP(0);

    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w1, 0, C * C * 2, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, C * 9 * C * 2, 0x00020000);

    // ---- staging maps.  x halo tile: one wave instruction = 1024 B = 1024 / ROWB rows; wave w stages pieces w, w + 8, ...
    unsigned h_off[G::XPW];
#pragma unroll
    for (int j = 0; j < G::XPW; ++j) {
        const int piece = wave + 8 * j;
        const int r = piece * (1024 / G::ROWB) + lane / G::NCH;
        const int hy = r / CP_HW, hx = r - hy * CP_HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const int lchunk = (lane % G::NCH) ^ cp_key<C>(r);
        const bool ok = r < CP_HALO && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        h_off[j] = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.XC + a.x_off + lchunk * 8) * 2) : OOR;
    }
    // sub-units: [C rows][64 k], 8 rows per wave instruction; wave w stages pieces w (+ 8)
    unsigned w1_off[G::PPW], w2_off[G::PPW];
#pragma unroll
    for (int j = 0; j < G::PPW; ++j) {
        const int row = (wave + 8 * j) * 8 + (lane >> 3);
        const int lchunk = (lane & 7) ^ ((row >> 1) & 7);
        w1_off[j] = (unsigned)((row * C + lchunk * 8) * 2);
        w2_off[j] = (unsigned)((row * 9 * C + lchunk * 8) * 2);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float b12_early = a.b12[tid & (2 * C - 1)];   // requested before every DMA: waiting for it later drains nothing
    __builtin_amdgcn_sched_barrier(0);
    auto dma_group = [&](int g) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < G::g_count(g); ++i) {
            const int q = G::g_first(g) + i;
            char *dst = smem + G::RING + ((g % G::NSLOT) * G::GS + i) * G::UNIT;
#pragma unroll
            for (int j = 0; j < G::PPW; ++j) {
                if (q < G::NW1)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_void *)(dst + (wave + 8 * j) * 1024), 16, (int)w1_off[j], q * 128, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_void *)(dst + (wave + 8 * j) * 1024), 16, (int)w2_off[j], (q - G::NW1) * 128, 0, 0);
            }
        }
    };
#pragma unroll
    for (int j = 0; j < G::XPW; ++j)
        if ((wave + 8 * j + 1) * (1024 / G::ROWB) <= G::BIAS_ROW)   // (the pieces of rows 184..191 are not staged: b1 | b2 live there)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(smem + G::T_OFF + (wave + 8 * j) * 1024), 16, (int)h_off[j], 0, 0, 0);
#pragma unroll
    for (int g = 0; g < G::NSLOT - 1 && g < G::NG; ++g) dma_group(g);
    float *bias12 = reinterpret_cast<float *>(smem + G::BIAS);
    if (tid < 2 * C) bias12[tid] = b12_early;

    char *T = smem + G::T_OFF;
    cp_f32x16 acc1[G::NFA];
#pragma unroll
    for (int j = 0; j < G::NFA; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
    cp_f32x16 acc2[G::NFB];
#pragma unroll
    for (int j = 0; j < G::NFB; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc2[j][e] = 0.f;
    int r0[G::NFB];   // halo row of this lane's pixel for tap (0, 0), per pixel fragment of phase B
#pragma unroll
    for (int j = 0; j < G::NFB; ++j) {
        const int p = 32 * (wq * G::NFB + j) + hp;
        r0[j] = (p >> 4) * CP_HW + (p & 15);
    }
    // this thread's 16-B pieces of the output: pixel e / NCH of the tile, chunk e % NCH, addressed through buffer descriptors (pixels outside
    // the image: out-of-range offset = read as zero / store dropped by the hardware, no divergent branch).  The shortcut values and the
    // pass-through channels are requested three sub-units before the end of phase B (their latency runs under its last MFMAs; earlier
    // they would only hold registers)
    __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);
    auto piece_off = [&](int it, int cstride, int c0) __attribute__((always_inline)) {
        const int e = tid + 512 * it, p = e / G::NCH, cc = e % G::NCH;
        const int yy = y0 + (p >> 4), xx = x0 + (p & 15);
        return (yy < a.H && xx < a.W) ? (unsigned)((((n * a.H + yy) * a.W + xx) * cstride + c0 + cc * 8) * 2) : OOR;
    };
    cp_u32x4 rres[G::EP_ITERS], pval[G::EP_ITERS];
#pragma unroll
    for (int it = 0; it < G::EP_ITERS; ++it) rres[it] = pval[it] = (cp_u32x4){0u, 0u, 0u, 0u};

#pragma unroll
    for (int g = 0; g < G::NG; ++g) {
        // this wave's pieces of group g (and, at g = 0, of the x tile) have landed; the younger groups stay in flight
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::younger(g)) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // ... everybody's have; and everybody is done with the slot of group g - 1
        __builtin_amdgcn_sched_barrier(0);
        if (g == 0) CP_STAMP(1);
        if (g == G::NW1 / G::GS) CP_STAMP(4);
        if (g + G::NSLOT - 1 < G::NG) dma_group(g + G::NSLOT - 1);
#pragma unroll
        for (int i = 0; i < G::g_count(g); ++i) {
            const int q = G::g_first(g) + i;
            const char *Wt = smem + G::RING + ((g % G::NSLOT) * G::GS + i) * G::UNIT;
            if (q < G::NW1) {
                // ---- phase A, K chunk q: T1 accumulators of halo-row fragments wq, wq + NWQ, ...
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const cp_bf16x8 fa = *reinterpret_cast<const cp_bf16x8 *>(Wt + cp_wswz(32 * wc + lr, 2 * kk + lh));
#pragma unroll
                    for (int j = 0; j < G::NFA; ++j) {
                        const int f = wq + G::NWQ * j;
                        if (f < 6) {
                            const cp_bf16x8 fb = *reinterpret_cast<const cp_bf16x8 *>(T + cp_swz<C>(32 * f + lr, 8 * q + 2 * kk + lh));
                            acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1[j], 0, 0, 0);
                        }
                    }
                }
                if (q == G::NW1 - 1) {
                    CP_STAMP(2);
                    // T1 -> T, over the x tile: bias, SiLU, zero outside the image, bf16; lane = halo row, 4 consecutive channels per group
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();   // every wave has read its x fragments
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < G::NFA; ++j) {
                        const int f = wq + G::NWQ * j;
                        if (f < 6) {
                            const int r = 32 * f + lr;
                            const int hy = r / CP_HW, hx = r - hy * CP_HW;
                            const bool ok = r < CP_HALO && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
#pragma unroll
                            for (int gg = 0; gg < 4; ++gg) {
                                const int c_local = 32 * wc + 8 * gg + 4 * lh;
                                const cp_f32x4 bv = *reinterpret_cast<const cp_f32x4 *>(bias12 + c_local);
                                cp_u32x2 pk;
                                pk.x = cp_pk_bf16(cp_silu(acc1[j][4 * gg + 0] + bv.x), cp_silu(acc1[j][4 * gg + 1] + bv.y));
                                pk.y = cp_pk_bf16(cp_silu(acc1[j][4 * gg + 2] + bv.z), cp_silu(acc1[j][4 * gg + 3] + bv.w));
                                if (!ok) pk.x = pk.y = 0u;
                                if (r < G::BIAS_ROW) *reinterpret_cast<cp_u32x2 *>(T + r * G::ROWB + (((4 * wc + gg) ^ cp_key<C>(r)) << 4) + 8 * lh) = pk;
                            }
                        }
                    }
                    CP_STAMP(3);
                    if (i + 1 < G::g_count(g)) {   // phase B starts inside this group: its reads wait for everybody's T1 rows
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // (otherwise the barrier at the top of the next group orders these writes before phase B's reads)
                }
            } else {
                // ---- phase B, sub-unit u = (64-channel chunk, tap)
                const int u = q - G::NW1, ch = u / 9, t = u % 9;
                const int shift = (t / 3) * CP_HW + (t % 3);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const cp_bf16x8 fa = *reinterpret_cast<const cp_bf16x8 *>(Wt + cp_wswz(32 * wc + lr, 2 * kk + lh));
#pragma unroll
                    for (int j = 0; j < G::NFB; ++j) {
                        const cp_bf16x8 fb = *reinterpret_cast<const cp_bf16x8 *>(T + cp_swz<C>(r0[j] + shift, 8 * ch + 2 * kk + lh));
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2[j], 0, 0, 0);
                    }
                }
                if (u == (G::NU > 3 ? G::NU - 3 : 0)) {
#pragma unroll
                    for (int it = 0; it < G::EP_ITERS; ++it) {
                        const unsigned xo = piece_off(it, a.XC, a.x_off);
                        if (a.shortcut) rres[it] = __builtin_bit_cast(cp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo, 0, 0));
                        if (a.pass) pval[it] = __builtin_bit_cast(cp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo, C * 2, 0));
                    }
                }
            }
        }
    }

    CP_STAMP(5);
    // ---- epilogue: bias, SiLU -> bf16 image [pixel][cout] over T, then 16-B pieces (+ shortcut) to y
    cp_f32x4 bv2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv2[g] = *reinterpret_cast<const cp_f32x4 *>(bias12 + C + 32 * wc + 8 * g + 4 * lh);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // every wave is done with T1 (and has b2 in registers)
    __builtin_amdgcn_sched_barrier(0);
    CP_STAMP(6);
    char *E = smem + G::T_OFF;
#pragma unroll
    for (int j = 0; j < G::NFB; ++j) {
        const int p_local = 32 * (wq * G::NFB + j) + hp;   // the pixel of accumulator column lr
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c_local = 32 * wc + 8 * g + 4 * lh;
            cp_u32x2 pk;
            pk.x = cp_pk_bf16(cp_silu(acc2[j][4 * g + 0] + bv2[g].x), cp_silu(acc2[j][4 * g + 1] + bv2[g].y));
            pk.y = cp_pk_bf16(cp_silu(acc2[j][4 * g + 2] + bv2[g].z), cp_silu(acc2[j][4 * g + 3] + bv2[g].w));
            *reinterpret_cast<cp_u32x2 *>(E + p_local * G::ES + c_local * 2) = pk;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    CP_STAMP(7);
#pragma unroll
    for (int it = 0; it < G::EP_ITERS; ++it) {
        const int e = tid + 512 * it, p = e / G::NCH, cc = e % G::NCH;
        cp_u32x4 v = *reinterpret_cast<const cp_u32x4 *>(E + p * G::ES + cc * 16);
        if (a.shortcut) {
            const cp_u32x4 rv = rres[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const cp_f32x2 sum = (cp_f32x2){__uint_as_float(v[k] << 16), __uint_as_float(v[k] & 0xffff0000u)} +
                                     (cp_f32x2){__uint_as_float(rv[k] << 16), __uint_as_float(rv[k] & 0xffff0000u)};
                v[k] = cp_pk_bf16(sum.x, sum.y);
            }
        }
        const unsigned yo = piece_off(it, a.YC, a.y_off);
        MD_BUFFER_STORE_B128(v, rs_y, yo, 0, 0);   // (store + hazard guard: aot.h)
        if (a.pass) MD_BUFFER_STORE_B128(pval[it], rs_y, yo, C * 2, 0);
    }
#ifdef MD_DIAG
    CP_STAMP(8);
    if (a.dbg && blockIdx.x == (gridDim.x / 2) && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stp[9] = __builtin_readcyclecounter();
        for (int i = 0; i < 12; ++i) a.dbg[i] = stp[i];
    }
#endif
}

// ---- C = 32 (YOLOv5s' first C3 block, 160 x 160 at the benchmark's shard): both weight matrices fit LDS whole (W1 2 KiB, W2 18 KiB), so there
// is no ring and no barrier inside the phases; 4 waves = the 4 pixel fragments of an 8 x 16 tile (phase A: halo-row fragments w and w + 4),
// one 32-cout fragment.  Weight rows are padded by 16 B (80 / 592 B: 16 consecutive rows then start in 16 different bank groups), T rows
// are 64 B with the chunk XOR-swizzled by (row >> 2) & 3.  K order = md_conv2d's for these layers ((tap, ci): korder 0).
constexpr int C32_TROW = 64, C32_W1ROW = 80, C32_W2ROW = 592;
constexpr int C32_T = 0, C32_W1 = CP_ROWS * C32_TROW, C32_W2 = C32_W1 + 32 * C32_W1ROW, C32_BIAS = C32_W2 + 32 * C32_W2ROW, C32_LDS = C32_BIAS + 256;
constexpr int C32_ES = 80;   // epilogue image row stride (over T: 128 x 80 B = 10 KiB)
__device__ __forceinline__ int c32_swz(int row, int chunk) { return row * C32_TROW + ((chunk ^ ((row >> 2) & 3)) << 4); }

__global__ __launch_bounds__(256, 4) void c3pair32_kernel(C3PairArgs a) {
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr unsigned OOR = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int hp = (int)(((0x73261540u >> ((lr >> 2) * 4)) & 7u) << 2) | (lr & 3);   // (see c3pair_body)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int pt = xcd * a.pt_per_xcd + slot;
    if (pt >= a.n_tiles) return;
    const int tx = pt % a.tiles_x, ty = (pt / a.tiles_x) % a.tiles_y, n = pt / (a.tiles_x * a.tiles_y);
    const int y0 = ty * CP_TH, x0 = tx * CP_TW;
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)a.y, 0, a.y_bytes, 0x00020000);

    // x halo tile by LDS-DMA: one wave instruction = 16 rows x 64 B; 12 pieces, wave w stages pieces w, w + 4, w + 8
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int piece = wave + 4 * j;
        const int r = piece * 16 + (lane >> 2);
        const int hy = r / CP_HW, hx = r - hy * CP_HW;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const int lchunk = (lane & 3) ^ ((r >> 2) & 3);
        const bool ok = r < CP_HALO && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const unsigned off = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.XC + a.x_off + lchunk * 8) * 2) : OOR;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(smem + C32_T + piece * 1024), 16, (int)off, 0, 0, 0);
    }
    // weights and biases: plain loads -> padded LDS rows (a.w1_ld / a.w2_ld: elements per packed row)
    {
        if (tid < 128) {
            const int row = tid >> 2, ch = tid & 3;
            const cp_u32x4 v = *reinterpret_cast<const cp_u32x4 *>(a.w1 + row * a.w1_ld + ch * 8);
            *reinterpret_cast<cp_u32x4 *>(smem + C32_W1 + row * C32_W1ROW + ch * 16) = v;
        }
#pragma unroll
        for (int it = 0; it < 5; ++it) {
            const int e = tid + 256 * it;
            if (e < 32 * 36) {
                const int row = e / 36, ch = e - row * 36;
                const cp_u32x4 v = *reinterpret_cast<const cp_u32x4 *>(a.w2 + row * a.w2_ld + ch * 8);
                *reinterpret_cast<cp_u32x4 *>(smem + C32_W2 + row * C32_W2ROW + ch * 16) = v;
            }
        }
        if (tid < 64) reinterpret_cast<float *>(smem + C32_BIAS)[tid] = a.b12[tid];
    }
    const float *bias12 = reinterpret_cast<const float *>(smem + C32_BIAS);
    char *T = smem + C32_T;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- phase A: T1 rows of fragments wave, wave + 4
    cp_f32x16 acc1[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc1[j][e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const cp_bf16x8 fa = *reinterpret_cast<const cp_bf16x8 *>(smem + C32_W1 + lr * C32_W1ROW + (2 * kk + lh) * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = wave + 4 * j;
            if (f < 6) {
                const cp_bf16x8 fb = *reinterpret_cast<const cp_bf16x8 *>(T + c32_swz(32 * f + lr, 2 * kk + lh));
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc1[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();   // every wave has read its x fragments
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int f = wave + 4 * j;
        if (f < 6) {
            const int r = 32 * f + lr;
            const int hy = r / CP_HW, hx = r - hy * CP_HW;
            const bool ok = r < CP_HALO && (unsigned)(y0 - 1 + hy) < (unsigned)a.H && (unsigned)(x0 - 1 + hx) < (unsigned)a.W;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const cp_f32x4 bv = *reinterpret_cast<const cp_f32x4 *>(bias12 + 8 * g + 4 * lh);
                cp_u32x2 pk;
                pk.x = cp_pk_bf16(cp_silu(acc1[j][4 * g + 0] + bv.x), cp_silu(acc1[j][4 * g + 1] + bv.y));
                pk.y = cp_pk_bf16(cp_silu(acc1[j][4 * g + 2] + bv.z), cp_silu(acc1[j][4 * g + 3] + bv.w));
                if (!ok) pk.x = pk.y = 0u;
                *reinterpret_cast<cp_u32x2 *>(T + r * C32_TROW + ((g ^ ((r >> 2) & 3)) << 4) + 8 * lh) = pk;
            }
        }
    }
    // the shortcut values and the pass-through channels of this thread's two output pieces (pixel e / 4, chunk e % 4)
    auto piece_off = [&](int it, int cstride, int c0) __attribute__((always_inline)) {
        const int e = tid + 256 * it, p = e >> 2, cc = e & 3;
        const int yy = y0 + (p >> 4), xx = x0 + (p & 15);
        return (yy < a.H && xx < a.W) ? (unsigned)((((n * a.H + yy) * a.W + xx) * cstride + c0 + cc * 8) * 2) : OOR;
    };
    cp_u32x4 rres[2], pval[2];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        rres[it] = pval[it] = (cp_u32x4){0u, 0u, 0u, 0u};
        const unsigned xo = piece_off(it, a.XC, a.x_off);
        if (a.shortcut) rres[it] = __builtin_bit_cast(cp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo, 0, 0));
        if (a.pass) pval[it] = __builtin_bit_cast(cp_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)xo, 64, 0));
    }
    __syncthreads();   // T1 complete

    // ---- phase B: pixel fragment = wave
    cp_f32x16 acc2;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
    const int pB = 32 * wave + hp;
    const int r0 = (pB >> 4) * CP_HW + (pB & 15);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int r = r0 + (t / 3) * CP_HW + (t % 3);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const cp_bf16x8 fa = *reinterpret_cast<const cp_bf16x8 *>(smem + C32_W2 + lr * C32_W2ROW + t * 64 + (2 * kk + lh) * 16);
            const cp_bf16x8 fb = *reinterpret_cast<const cp_bf16x8 *>(T + c32_swz(r, 2 * kk + lh));
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0);
        }
    }
    cp_f32x4 bv2[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv2[g] = *reinterpret_cast<const cp_f32x4 *>(bias12 + 32 + 8 * g + 4 * lh);
    __syncthreads();   // every wave is done with T1
    char *E = smem + C32_T;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        cp_u32x2 pk;
        pk.x = cp_pk_bf16(cp_silu(acc2[4 * g + 0] + bv2[g].x), cp_silu(acc2[4 * g + 1] + bv2[g].y));
        pk.y = cp_pk_bf16(cp_silu(acc2[4 * g + 2] + bv2[g].z), cp_silu(acc2[4 * g + 3] + bv2[g].w));
        *reinterpret_cast<cp_u32x2 *>(E + pB * C32_ES + (8 * g + 4 * lh) * 2) = pk;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int e = tid + 256 * it, p = e >> 2, cc = e & 3;
        cp_u32x4 v = *reinterpret_cast<const cp_u32x4 *>(E + p * C32_ES + cc * 16);
        if (a.shortcut) {
            const cp_u32x4 rv = rres[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const cp_f32x2 sum = (cp_f32x2){__uint_as_float(v[k] << 16), __uint_as_float(v[k] & 0xffff0000u)} +
                                     (cp_f32x2){__uint_as_float(rv[k] << 16), __uint_as_float(rv[k] & 0xffff0000u)};
                v[k] = cp_pk_bf16(sum.x, sum.y);
            }
        }
        const unsigned yo = piece_off(it, a.YC, a.y_off);
        MD_BUFFER_STORE_B128(v, rs_y, yo, 0, 0);
        if (a.pass) MD_BUFFER_STORE_B128(pval[it], rs_y, yo, 64, 0);
    }
}

// (two kernels instead of one template: the register budget differs -- six / four waves per SIMD -- and hipcc's host pass rejects launch
// bounds that depend on a template parameter)
__global__ __launch_bounds__(512, 6) void c3pair64_kernel(C3PairArgs a) { c3pair_body<64>(a); }
__global__ __launch_bounds__(512, 4) void c3pair128_kernel(C3PairArgs a) { c3pair_body<128>(a); }

}  // namespace md

using namespace md;

#ifdef MD_DIAG
static unsigned long long *g_c3_stamp_buf = nullptr;
extern "C" int md_diag_set_c3_stamp_buffer(void *p) { g_c3_stamp_buf = (unsigned long long *)p; return MD_OK; }
#endif

// in : x[N,H,W,XC] bf16, w1[C,C] bf16, b12[2 C] f32 (= b1 | b2), w2[C, 9 C] bf16 (K = (ci / 64) * 576 + tap * 64 + ci % 64: md_conv2d's
//      korder 1; for C = 64 that is tap * 64 + ci; C = 32: [32, 64] and [32, 320] with K = tap * 32 + ci, korder 0), BN folded, as md_conv2d packs
//      them;  C = 32, 64 or 128
// out: y[N,H,W,YC] bf16 -- another buffer than x (a workgroup's halo pixels are other workgroups' outputs)
// extra: md_c3_pair_attrs (required)
extern "C" int md_c3_pair(MD_AOT_ARGS) {
    if (nparam != 5) return MD_ERR_NPARAM;
    if (!params || !ndims || !shapes || !extra) return MD_ERR_ARG;
    for (int i : {0, 1, 3, 4})
        if (!dtype_is(dtypes, i, "bfloat16")) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 2, "float32")) return MD_ERR_ARG;
    if (ndims[0] != 4 || ndims[4] != 4 || ndims[1] != 2 || ndims[3] != 2) return MD_ERR_ARG;
    const md_c3_pair_attrs *at = (const md_c3_pair_attrs *)extra;
    const int64_t N = shapes[0][0], H = shapes[0][1], W = shapes[0][2], XC = shapes[0][3], YC = shapes[4][3];
    const int64_t C = shapes[1][0];
    const int64_t k1 = (C + 63) / 64 * 64, k2 = (9 * C + 63) / 64 * 64;   // md_conv2d pads the packed K to a multiple of 64
    if ((C != 32 && C != 64 && C != 128) || shapes[1][1] != k1 || shapes[3][0] != C || shapes[3][1] != k2 || numel(ndims, shapes, 2) != 2 * C) return MD_ERR_ARG;
    if (shapes[4][0] != N || shapes[4][1] != H || shapes[4][2] != W) return MD_ERR_ARG;
    const int64_t span = at->pass_through ? 2 * C : C;
    if (at->x_c_off < 0 || at->y_c_off < 0 || at->x_c_off % 8 || at->y_c_off % 8 || XC % 8 || YC % 8 || at->x_c_off + span > XC || at->y_c_off + span > YC)
        return MD_ERR_ARG;
    if (N * H * W == 0) return MD_OK;
    for (int i = 0; i < 5; ++i)
        if (!params[i]) return MD_ERR_ARG;
    {   // x and y must not overlap (see above)
        const char *xb = (const char *)params[0], *yb = (const char *)params[4];
        const long long xn = N * H * W * XC * 2, yn = N * H * W * YC * 2;
        if (xb < yb + yn && yb < xb + xn) return MD_ERR_ARG;
    }
    if (H > 32000 || W > 32000) return MD_ERR_SIZE;
    if (N * H * W * XC * 2 >= 0x7fff0000LL || N * H * W * YC * 2 >= 0x7fff0000LL) return MD_ERR_SIZE;   // 32-bit buffer offsets into x and y
    C3PairArgs a;
    a.x = (const uint16_t *)params[0]; a.w1 = (const uint16_t *)params[1]; a.b12 = (const float *)params[2];
    a.w2 = (const uint16_t *)params[3]; a.y = (uint16_t *)params[4];
    a.N = (int)N; a.H = (int)H; a.W = (int)W; a.XC = (int)XC; a.YC = (int)YC;
    a.x_off = at->x_c_off; a.y_off = at->y_c_off; a.shortcut = at->shortcut ? 1 : 0; a.pass = at->pass_through ? 1 : 0;
    a.tiles_x = (int)((W + CP_TW - 1) / CP_TW); a.tiles_y = (int)((H + CP_TH - 1) / CP_TH);
    const long long n_tiles = N * a.tiles_x * a.tiles_y;
    if (n_tiles > 0x7fffffffLL / 8) return MD_ERR_SIZE;
    a.n_tiles = (int)n_tiles;
    a.pt_per_xcd = (a.n_tiles + 7) / 8;
    a.x_bytes = (unsigned)(N * H * W * XC * 2);
    a.y_bytes = (unsigned)(N * H * W * YC * 2);
    a.w1_ld = (int)k1; a.w2_ld = (int)k2;
    a.dbg = nullptr;
#ifdef MD_DIAG
    a.dbg = g_c3_stamp_buf;
#endif
    void (*k)(C3PairArgs) = C == 32 ? c3pair32_kernel : (C == 64 ? c3pair64_kernel : c3pair128_kernel);
    const int lds = C == 32 ? C32_LDS : (C == 64 ? CpGeom<64>::LDS : CpGeom<128>::LDS);
    if (ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3((unsigned)(a.pt_per_xcd * 8)), dim3(C == 32 ? 256 : 512), lds, (hipStream_t)stream, a);
    md_note_conv_kernel(MD_CONV_KERNEL_C3_PAIR);
    MD_HIP_TRY(hipGetLastError());
    return MD_OK;
}
