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
// Four kernels (dispatcher + cost model at the end of the file):
//   conv_pingpong_kernel  256 x 256 x 64 tile, one 8-wave workgroup per CU, two staggered wave groups, LDS-DMA stream
//                         with counted vmcnt -- the MFMA-bound layers (K >= 1024, Cout % 256 == 0), optionally with a
//                         fused 1x1 head out of the epilogue image (md_conv2d_head), in a PERSISTENT form for the long-K layers
//                         without residual (next tile's prologue DMAs under a barrier-free slab epilogue), and with its K tiles
//                         past the first tensor staged from a second, strided one (the long-K md_conv1x1_dual GEMMs);
//   conv1x1_stream_kernel weight-stationary pointwise layers with K <= 512: weights in registers, activations through an LDS-DMA
//                         ring, residual DMAed into the wave's epilogue image -- the HBM-bound expand / lateral convs;
//   conv_igemm_kernel     128 x 128 x 64 (or 64- / 32-cout) tile, 4 waves, ONE LDS staging buffer -> four resident
//                         workgroups per CU -- everything else;
//   conv3x3_halo_kernel   3x3 / s1 layers with Cout <= 64: the 10 x 18 halo staged once per 64-channel chunk.
// Common: v_mfma_f32_32x32x16_bf16; LDS tiles [row][64 k] bf16 (128-B rows) with the 16-B chunk index XOR-swizzled by
// (row>>1)&7 (applied to the DMA's SOURCE chunk and to the ds_read_b128 fragment reads: conflict free); epilogue = bias
// from LDS (requested at kernel start) -> packed adds -> v_cvt_pk_bf16_f32 -> packed ReLU -> LDS transpose -> +residual
// -> non-temporal 16-B stores.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aot.h"

namespace md {

// MD_DIAG (tools/ only: `make -C minddet_amd/csrc diag` -> libminddet_hip_diag.so): the timing ablations and in-kernel stamps of
// tools/pp_stamps.py / igemm_stamps.py.  They do not compute the convolution, so the PRODUCT library neither contains them
// nor accepts their variant ids (md_conv2d returns MD_ERR_ARG); stamps go to a buffer of their own (md_diag_set_stamp_buffer),
// never into an output tensor.
#ifdef MD_DIAG
constexpr bool kDiag = true;
#else
constexpr bool kDiag = false;
#endif

typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

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
    unsigned x_bytes, w_bytes;  // buffer sizes for the LDS-DMA descriptors (< 2 GiB each)
    // generalised addressing (sub-pixel transposed conv, channel-concat outputs); plain conv: pad_top = pad_left
    // = pad, (Ho,Wo) is the full output, os = 1, oy = ox = c_off = 0, Ctot = Cout
    int pad_top, pad_left;  // input window origin = (ho*stride - pad_top, wo*stride - pad_left)
    int Hf, Wf, Ctot;       // full output tensor dims [N,Hf,Wf,Ctot]
    int os, oy, ox, c_off;  // output pixel (ho*os + oy, wo*os + ox), channels [c_off, c_off + Cout)
    int adv;                // 0: plain (off = m*Cout + c)
    int korder;             // 0: K = (tap, ci); 1: K = (ci/64, tap, ci%64)  (MODE 2 only)
    int single_buf;         // LDS-DMA K loop with ONE staging buffer (four resident workgroups per CU)
    const uint16_t *w2;     // fused 1x1 head (conv_pingpong_kernel<.., HEAD>): [>=16][256] bf16, K = the conv's 256 output channels
    const float *b2;        // [>=16] fp32
    uint16_t *y2;           // [N,Ho,Wo,16] bf16 -- the ONLY tensor a HEAD launch writes
    int pointwise;          // 1x1 / stride 1 / pad 0 (input pixel index == output pixel index)
    int bias_lds_off;       // byte offset of the CT-float bias copy in LDS (past the staging buffers and the epilogue image)
    int stamp;              // MD_DIAG builds only (variant 25): a mid-grid workgroup writes its cycle stamps to a.dbg
    unsigned long long *dbg; // MD_DIAG builds only: the stamp buffer (md_diag_set_stamp_buffer), never an output tensor
    int res_up;             // 1: residual is [N, ceil(Ho/2), ceil(Wo/2), Cout], read with nearest 2x upsampling
                            //    (the FPN top-down add fused into the lateral 1x1 conv); plain addressing only
    int Xs;                 // pixel stride of x in channels (== Cin unless the input is a channel slice of a wider tensor;
                            // a.x then already points at the slice's first channel)
    // DUAL (md_conv1x1_dual): K tiles >= nk_a come from a second tensor x2 [N,H2,W2,Xs2] read at (ho * stride2, wo * stride2)
    const uint16_t *x2;
    unsigned x2_bytes;
    int H2, W2, Xs2, stride2, nk_a;
    int tune;               // conv1x1_stream_kernel cache-policy bits (md_conv_tune.stream_cache_bits): 1 = x DMA nt, 2 = residual DMA nt, 4 = stores nt
    int tiles_x, tiles_y;   // HALO form of the ping-pong kernel: 16 x 16-pixel tiles per image row / rows of such tiles
    int tiles_strip;        // HALO: 8 x 32-pixel tiles of the bottom strip (image rows 16 * tiles_y ..; 0 = none), tiles per image = tiles_x * tiles_y + tiles_strip
    int Rs;                 // 0: the residual has the output's layout; > 0: residual pixel m, channel c at m*Rs + c (a channel
                            // slice of a wider [N,Ho,Wo,Rs] tensor, a.res pointing at its first channel)
};

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((unsigned)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) {
    // round-to-nearest-even; NaN stays NaN (quiet)
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// two fp32 -> packed bf16 (round-to-nearest-even) in ONE instruction (gfx950 v_cvt_pk_bf16_f32).  r01 stamps: the software
// rounding (9 VALU per value, 64 values per lane) made the epilogue VALU-bound at four workgroups per CU.
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ unsigned pk_relu_bf16(unsigned v) {  // max(x, 0) on a packed bf16 pair: as signed 16-bit integers
    unsigned r;
    asm("v_pk_max_i16 %0, %1, 0" : "=v"(r) : "v"(v));
    return r;
}
typedef float f32x2 __attribute__((ext_vector_type(2)));

// SiLU with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (v_div_scale / fmas / fixup: ~10 VALU per
// value, 64 values per lane in an epilogue that is VALU-bound at four workgroups per CU); the result is rounded to bf16.
__device__ __forceinline__ float silu(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

constexpr int BK = 64;
constexpr int ROWB = BK * 2;  // bytes per LDS tile row

__device__ __forceinline__ int swz(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 1) & 7)) << 4); }

// id of the kernel the dispatcher chose for this host thread's last md_conv2d call (md_conv2d_last_kernel): lets a
// profiler-less caller (bench.py) attribute per-launch timings to kernels
static thread_local int g_last_kernel = 0;
// kernels launched so far by this host thread's conv-family calls (md_conv2d_launch_count): a call on a batch past the chunk limit launches once per image chunk
static thread_local long long g_launch_count = 0;
// Per-call tuning (md_conv_tune of the op's attribute struct, resolved against the defaults): the library keeps no mutable knob.
struct Tune {
    long long chunk_limit;   // activation bytes above which the batch is sliced into image chunks (the kernels' 32-bit DMA offsets)
    int stream_rounds, stream_wgs_per_cu, stream_cache_bits, pers_min_k, dual_pp_min_k;
    bool stream_narrow;      // stream_cache_bits & 16: the K = 512 form on 4-wave (128-cout) workgroups also where the 8-wave form applies (A/B)
};
static Tune resolve_tune(const md_conv_tune *t) {
    Tune r = {0x7fff0000LL, 1, 2, 6, 2304, 768, false};
    if (!t) return r;
    r.stream_narrow = (t->stream_cache_bits & 16) != 0;
    if (t->chunk_limit > 0 && t->chunk_limit < 0x7fff0000) r.chunk_limit = t->chunk_limit;
    if (t->stream_rounds >= 1 && t->stream_rounds <= 64) r.stream_rounds = t->stream_rounds;
    if (t->stream_wgs_per_cu >= 1 && t->stream_wgs_per_cu <= 8) r.stream_wgs_per_cu = t->stream_wgs_per_cu;
    if (t->stream_cache_bits & 8) r.stream_cache_bits = t->stream_cache_bits & 7;
    if (t->pers_min_k >= 1) r.pers_min_k = t->pers_min_k;
    if (t->dual_pp_min_k >= 128) r.dual_pp_min_k = t->dual_pp_min_k;
    return r;
}


// 16 zero bytes: the source of every out-of-image / past-K chunk when staging with LDS-DMA
__device__ __attribute__((aligned(16))) unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// Template: NT threads (256 = 4 waves, 512 = 8 waves), waves arranged WC (cout) x WP (pixels), each wave
// owning FC x FP accumulator tiles of 32x32.
// MODE 0: register-staged tiles (global_load -> VGPR -> ds_write), 64-bit addressing, any size.
// MODE 1: LDS-DMA staging (buffer_load_dwordx4 ... lds: no VGPR round trip, hardware range check = free
//         zero fill of padding taps), generic K walk (any Cin % 8 == 0, e.g. the 7x7 stem on 8 channels).
// MODE 2: LDS-DMA, Cin % 64 == 0 and <= 32 taps: one K tile never straddles a tap, so the tap walk is
//         scalar (SALU) and goes into the instruction's soffset; per tile a lane spends 3 VALU per
//         activation row (tap-validity bit -> select the out-of-range offset) and none on weights.
// The LDS image of a DMA is lane-linear per wave instruction (8 rows x 128 B), so the XOR swizzle is
// applied to the per-lane SOURCE chunk and to the fragment reads, never to the destination.
// K loop: ONE LDS staging buffer by default (a.single_buf; 34 KiB -> four resident workgroups per CU whose serial
// DMA -> MFMA -> epilogue chains overlap each other), or two with the next tile's DMA issued before the MFMAs (variant 2).
// Measured and removed in r01: 3-stage ring with counted vmcnt, producer/consumer waves, 4-stage
// BK-32 ring, DMA issue interleaved with the MFMA clusters, 256x256 tile on this loop.
// GEN 1: the general epilogue (sub-pixel output addressing, upsampled residual, any activation).  GEN 0 / 2: the plain
// instantiations for ReLU-or-none / SiLU layers whose output is a whole tensor or a channel range of a concat buffer
// (offset = m * Ctot + c_off + c): the division-heavy address code of the general epilogue is compiled out (it was 2/3 of the
// kernel's instructions) and the activation is a compile-time choice.
// DUAL 1 (MODE 2, 1x1): the K axis is the concatenation of TWO input tensors -- K tiles < a.nk_a from x (pixel = output pixel), the rest
// from x2 sampled with stride a.stride2 (md_conv1x1_dual: a bottleneck's expand conv and its strided downsample conv as ONE GEMM).
template <int NT, int WC, int WP, int FC, int FP, int MODE, int GEN = 1, int DUAL = 0>
__global__ __launch_bounds__(NT, NT == 256 ? 3 : 2) void conv_igemm_kernel(ConvArgs a) {
    constexpr bool GLDS = MODE != 0;
    constexpr int CT = WC * FC * 32, PT = WP * FP * 32;
    constexpr int RPP = NT / 8;                            // tile rows staged per pass of the workgroup
    constexpr int A_ROWS = CT / RPP, B_ROWS = PT / RPP;    // 16-B chunks per thread per tile
    constexpr int TILE_BYTES = (CT + PT) * ROWB;
    constexpr int EP_STRIDE = CT * 2 + 16;                 // epilogue image row stride (bytes)
    constexpr unsigned OOR = 0x80000000u;                  // byte offset past any buffer (< 2 GiB): reads as 0
    static_assert(WC * WP * 64 == NT, "wave grid must cover the workgroup");
    static_assert(CT % RPP == 0 && PT % RPP == 0, "tile rows must be a multiple of the staging pass");
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
    unsigned long long stp[6] = {0, 0, 0, 0, 0, 0};
    if (kDiag && a.stamp) stp[0] = __builtin_readcyclecounter();
    // The tile's bias is requested NOW and parked in LDS for the epilogue.  r01 stamps (tools/igemm_stamps.py): fetched
    // after the K loop, this one dependent global load exposed 9-12k cycles of loaded-HBM latency in every workgroup.
    const float bias_early = tid < CT ? a.bias[cout0 + tid] : 0.f;

    // this thread stages rows row0 + RPP*i; physical 16-B slot (tid & 7) of the row holds LOGICAL k-chunk
    // `chunk` (the swizzle term (row>>1)&7 is the same for all of a thread's rows because RPP % 16 == 0)
    const int row0 = tid >> 3;
    const int chunk = GLDS ? ((tid & 7) ^ ((row0 >> 1) & 7)) : (tid & 7);
    const int n_taps = a.kh * a.kw;
    const int nk = a.Kpad / BK;

    // per-thread pixel rows of the B (activation) tile
    int p_hw0[B_ROWS];        // packed (hi0 << 16) | (wi0 & 0xffff): top-left input coord of the window
    int p_base[B_ROWS];       // MODE 0: n*H*W (pixel index) or -1 past M.  MODE 1/2: byte offset of (n,hi0,wi0,chunk)
    unsigned p_taps[B_ROWS];  // MODE 2: bit t set <=> tap t of this row is inside the image (0 past M)
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
        const int m = pix0 + row0 + RPP * i;
        p_hw0[i] = 0; p_base[i] = -1; p_taps[i] = 0u;
        if (MODE == 2 && a.pointwise) {
            // 1x1 / stride 1 / pad 0: input pixel == output pixel, no (n, ho, wo) decode (two integer divisions per row:
            // on the short-K layers this setup cost as many VALU cycles as the whole K loop)
            if (m < a.M) { p_base[i] = m * a.Xs * 2 + chunk * 16; p_taps[i] = 1u; }
        } else if (m < a.M) {
            const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
            const int ho = r / a.Wo, wo = r - ho * a.Wo;
            const int hi0 = ho * a.stride - a.pad_top, wi0 = wo * a.stride - a.pad_left;
            p_hw0[i] = (hi0 << 16) | (wi0 & 0xffff);
            if constexpr (MODE == 0) p_base[i] = n * a.H * a.W;
            else p_base[i] = (((n * a.H + hi0) * a.W + wi0) * a.Xs) * 2 + (MODE == 2 ? chunk * 16 : 0);
            if constexpr (MODE == 2) {
                unsigned bits = 0u, bit = 1u;
                for (int dy = 0; dy < a.kh; ++dy)
                    for (int dx = 0; dx < a.kw; ++dx, bit <<= 1)
                        if ((unsigned)(hi0 + dy) < (unsigned)a.H && (unsigned)(wi0 + dx) < (unsigned)a.W) bits |= bit;
                p_taps[i] = bits;
            }
        } else if constexpr (MODE != 0) p_hw0[i] = (int)0x80008000;  // hi0 = wi0 = -32768: every tap out of image
    }
    // generic K walk (MODE 0/1): tap index and channel-chunk of this thread's chunk, advanced 8 chunks per tile
    int q_tap = chunk / a.cpt, q_cc = chunk - q_tap * a.cpt;
    int q_kh = q_tap / a.kw, q_kw = q_tap - q_kh * a.kw;
    auto advance_k = [&]() {
        q_cc += 8;
        while (q_cc >= a.cpt) {
            q_cc -= a.cpt;
            ++q_tap;
            if (++q_kw == a.kw) { q_kw = 0; ++q_kh; }
        }
    };
    // scalar K walk (MODE 2): the whole tile sits in tap s_tap at channel offset s_cc0 (multiples of 8 chunks)
    int s_tap = 0, s_cc0 = 0, s_kh = 0, s_kw = 0;

    u32x4 ra[GLDS ? 1 : A_ROWS], rb[GLDS ? 1 : B_ROWS];
    // ---- staging, register path (MODE 0)
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            ra[GLDS ? 0 : i] = *reinterpret_cast<const u32x4 *>(a.w + (size_t)(cout0 + row0 + RPP * i) * a.Kpad + kt * BK + chunk * 8);
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            const int hi = (p_hw0[i] >> 16) + q_kh, wi = (int)(short)(p_hw0[i] & 0xffff) + q_kw;
            const bool ok = q_tap < n_taps && p_base[i] >= 0 && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4 *>(a.x + ((size_t)(p_base[i] + hi * a.W + wi)) * a.Xs + q_cc * 8);
            rb[GLDS ? 0 : i] = v;
        }
        advance_k();
    };
    auto store_tile = [&](int buf) {
        char *A = smem + buf * TILE_BYTES, *B = A + CT * ROWB;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<u32x4 *>(A + swz(row0 + RPP * i, tid & 7)) = ra[GLDS ? 0 : i];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<u32x4 *>(B + swz(row0 + RPP * i, tid & 7)) = rb[GLDS ? 0 : i];
    };
    // ---- staging, LDS-DMA path: one wave instruction fills 8 rows x 128 B = 1 KiB, destination lane-linear
    __amdgpu_buffer_rsrc_t rs_w, rs_x;
    int a_off0 = 0;
    if constexpr (GLDS) {
        rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
        rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
        a_off0 = ((cout0 + row0) * a.Kpad + chunk * 8) * 2;
    }
    const int wrow = wave * 8;  // first row of this wave's 8-row group inside a staging pass
    __amdgpu_buffer_rsrc_t rs_x2 = rs_x;
    unsigned p_base2[DUAL ? B_ROWS : 1];
    if constexpr (DUAL != 0) {
        rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.x2, 0, a.x2_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            const int m = pix0 + row0 + RPP * i;
            p_base2[i] = OOR;
            if (m < a.M) {
                const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
                const int ho = r / a.Wo, wo = r - ho * a.Wo;
                p_base2[i] = (unsigned)((((n * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Xs2) * 2 + chunk * 16);
            }
        }
    }
    auto dma_tile = [&](int kt, int buf) {
        typedef __attribute__((address_space(3))) void lds_void;
        char *A = smem + buf * TILE_BYTES, *B = A + CT * ROWB;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(A + (wrow + RPP * i) * ROWB), 16,
                                                     a_off0 + i * (RPP * a.Kpad * 2), kt * (BK * 2), 0, 0);
        if constexpr (DUAL != 0) {
            if (kt >= a.nk_a) {   // second tensor: 64-channel chunk kt - nk_a of the strided pixel
#pragma unroll
                for (int i = 0; i < B_ROWS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void *)(B + (wrow + RPP * i) * ROWB), 16, (int)p_base2[i],
                                                             (kt - a.nk_a) * (BK * 2), 0, 0);
                return;
            }
        }
        if constexpr (MODE == 2) {
            const int soff = ((s_kh * a.W + s_kw) * a.Xs + s_cc0 * 8) * 2;
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) {
                const unsigned voff = ((p_taps[i] >> s_tap) & 1u) ? (unsigned)(p_base[i] + soff) : OOR;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(B + (wrow + RPP * i) * ROWB), 16, (int)voff, 0, 0, 0);
            }
            if (a.korder == 0) {  // K ordered (tap, ci): channel chunks fastest
                s_cc0 += 8;
                if (s_cc0 == a.cpt) {
                    s_cc0 = 0; ++s_tap;
                    if (++s_kw == a.kw) { s_kw = 0; ++s_kh; }
                }
            } else {              // K ordered (ci chunk, tap, ci in chunk): taps fastest -> consecutive tiles
                ++s_tap;          // re-read almost the same activation lines (L1/L2 reuse of the 3x3 window)
                if (++s_kw == a.kw) { s_kw = 0; ++s_kh; }
                if (s_tap == n_taps) { s_tap = 0; s_kh = 0; s_kw = 0; s_cc0 += 8; }
            }
        } else {
            const int tapoff = ((q_kh * a.W + q_kw) * a.Xs + q_cc * 8) * 2;
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) {
                const int hi = (p_hw0[i] >> 16) + q_kh, wi = (int)(short)(p_hw0[i] & 0xffff) + q_kw;
                const bool ok = q_tap < n_taps && (unsigned)hi < (unsigned)a.H && (unsigned)wi < (unsigned)a.W;
                const unsigned voff = ok ? (unsigned)(p_base[i] + tapoff) : OOR;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(B + (wrow + RPP * i) * ROWB), 16, (int)voff, 0, 0, 0);
            }
            advance_k();
        }
    };

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    // fragment read addresses: the swizzle term is the same for every fragment row of a lane (rows differ by
    // multiples of 32), so 4 bases per operand (one per k-step) + immediates cover a whole tile
    int fa_off[BK / 16], fb_off[BK / 16];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
        fa_off[kk] = swz(wc * FC * 32 + lr, kk * 2 + lh);
        fb_off[kk] = CT * ROWB + swz(wp * FP * 32 + lr, kk * 2 + lh);
    }
    auto compute_tile32 = [&](int buf) {
        const char *T = smem + buf * TILE_BYTES;
        // fragments of k-step kk+1 are fetched while the MFMAs of k-step kk run (two register sets)
        bf16x8 fa[2][FC], fb[2][FP];
#pragma unroll
        for (int i = 0; i < FC; ++i) fa[0][i] = *reinterpret_cast<const bf16x8 *>(T + fa_off[0] + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < FP; ++j) fb[0][j] = *reinterpret_cast<const bf16x8 *>(T + fb_off[0] + j * 32 * ROWB);
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            if (kk + 1 < BK / 16) {
#pragma unroll
                for (int i = 0; i < FC; ++i) fa[(kk + 1) & 1][i] = *reinterpret_cast<const bf16x8 *>(T + fa_off[kk + 1] + i * 32 * ROWB);
#pragma unroll
                for (int j = 0; j < FP; ++j) fb[(kk + 1) & 1][j] = *reinterpret_cast<const bf16x8 *>(T + fb_off[kk + 1] + j * 32 * ROWB);
            }
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[kk & 1][i], fb[kk & 1][j], acc[i][j], 0, 0, 0);
        }
    };

    auto compute_tile = [&](int buf) { compute_tile32(buf); };

    if constexpr (GLDS) {
        if (a.single_buf) {
            // short-K, HBM/latency-bound layers: half the LDS -> twice the resident workgroups, whose epilogues
            // (residual loads, stores) then overlap each other's loops
            for (int kt = 0; kt < nk; ++kt) {
                dma_tile(kt, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (kDiag && a.stamp && kt == 0) stp[1] = __builtin_readcyclecounter();
                compute_tile(0);
                __syncthreads();
            }
            if (kDiag && a.stamp) stp[2] = __builtin_readcyclecounter();
        } else {
        dma_tile(0, 0);
        for (int kt = 0; kt < nk; ++kt) {
            // tile kt has landed (every wave waits for its own DMAs, the barrier publishes them) and every
            // wave has finished reading the other buffer in the previous iteration -> safe to refill it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nk) dma_tile(kt + 1, (kt + 1) & 1);
            compute_tile(kt & 1);
        }
        __syncthreads();
        }
    } else {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) load_tile(kt + 1);
            compute_tile(kt & 1);
            if (kt + 1 < nk) store_tile((kt & 1) ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue.  Residual rows are fetched first so that their HBM latency overlaps the LDS transpose.
    constexpr int CPP = CT / 8;                 // 16-B chunks per pixel row of the tile
    constexpr int EP_ITERS = PT * CPP / NT;     // chunks per thread
    auto out_offset = [&](int m, int c) -> size_t {
        if (!(GEN == 1 && a.adv)) return (size_t)m * a.Ctot + a.c_off + c;   // plain, or a channel-concat output
        const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
        const int ho = r / a.Wo, wo = r - ho * a.Wo;
        return (((size_t)n * a.Hf + ho * a.os + a.oy) * a.Wf + wo * a.os + a.ox) * a.Ctot + a.c_off + c;
    };
    u32x4 rres[EP_ITERS];
    if (GEN == 1 && a.res_up) {
        // upsampled residual (FPN top-down add): (n, ho, wo) of the thread's first pixel by division, the following
        // pixels (NT / CPP apart) by carry -- two integer divisions per thread instead of two per 16-B chunk
        const int c = cout0 + (tid % CPP) * 8;
        int m = pix0 + tid / CPP;
        int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
        int ho = r / a.Wo, wo = r - ho * a.Wo;
        const int Hr = (a.Ho + 1) >> 1, Wr = (a.Wo + 1) >> 1;
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            rres[it] = (u32x4){0u, 0u, 0u, 0u};
            if (m < a.M && c < a.Cout)
                rres[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(
                    a.res + (((size_t)n * Hr + (ho >> 1)) * Wr + (wo >> 1)) * a.Cout + c));
            m += NT / CPP; wo += NT / CPP;
            while (wo >= a.Wo) { wo -= a.Wo; if (++ho == a.Ho) { ho = 0; ++n; } }
        }
    } else if (a.res) {
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int e = tid + it * NT;
            const int p_local = e / CPP, cc = e % CPP;
            const int m = pix0 + p_local, c = cout0 + cc * 8;
            rres[it] = (u32x4){0u, 0u, 0u, 0u};
            if (m < a.M && c < a.Cout)
                rres[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.res + (a.Rs ? (size_t)m * a.Rs + c : out_offset(m, c))));
        }
    }
    // bias (+ReLU when no residual) -> bf16x4 -> LDS [pixel][cout] image
    char *E = smem;
    float *bias_lds = reinterpret_cast<float *>(smem + a.bias_lds_off);
    if (tid < CT) bias_lds[tid] = bias_early;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FC; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c_local = (wc * FC + i) * 32 + 8 * g + 4 * lh;  // 4 consecutive couts
            const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + c_local);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int p_local = (wp * FP + j) * 32 + lr;
                // packed adds (v_pk_add_f32), one-instruction bf16 pack, ReLU on the packed pair: 8 VALU per 4 values
                f32x2 s01 = (f32x2){acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]} + (f32x2){bv.x, bv.y};
                f32x2 s23 = (f32x2){acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} + (f32x2){bv.z, bv.w};
                if (GEN == 2 || (GEN == 1 && a.relu == 2)) { s01.x = silu(s01.x); s01.y = silu(s01.y); s23.x = silu(s23.x); s23.y = silu(s23.y); }
                uint2 pk;
                pk.x = pk_bf16(s01.x, s01.y);
                pk.y = pk_bf16(s23.x, s23.y);
                if (a.relu == 1 && !a.res) { pk.x = pk_relu_bf16(pk.x); pk.y = pk_relu_bf16(pk.y); }
                *reinterpret_cast<uint2 *>(E + p_local * EP_STRIDE + c_local * 2) = pk;
            }
        }
    }
    if (kDiag && a.stamp) stp[3] = __builtin_readcyclecounter();
    __syncthreads();
    if (kDiag && a.stamp) stp[4] = __builtin_readcyclecounter();
    // ---- coalesced NHWC store: 16 B (8 couts) per lane, CT/8 lanes per pixel
#pragma unroll
    for (int it = 0; it < EP_ITERS; ++it) {
        const int e = tid + it * NT;
        const int p_local = e / CPP, cc = e % CPP;
        const int m = pix0 + p_local, c = cout0 + cc * 8;
        if (m >= a.M || c >= a.Cout) continue;
        u32x4 v = *reinterpret_cast<const u32x4 *>(E + p_local * EP_STRIDE + cc * 16);
        const size_t off = out_offset(m, c);
        if (a.res) {
            const u32x4 rv = rres[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 sum = (f32x2){__uint_as_float(v[k] << 16), __uint_as_float(v[k] & 0xffff0000u)} +
                                  (f32x2){__uint_as_float(rv[k] << 16), __uint_as_float(rv[k] & 0xffff0000u)};
                v[k] = pk_bf16(sum.x, sum.y);
                if (a.relu == 1) v[k] = pk_relu_bf16(v[k]);
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(a.y + off));
    }
    if (kDiag && a.stamp && a.dbg && blockIdx.x == gridDim.x / 2) {
        stp[5] = __builtin_readcyclecounter();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t_done = __builtin_readcyclecounter();
        if (tid == 0) {
            for (int i = 0; i < 6; ++i) a.dbg[i] = stp[i];
            a.dbg[6] = t_done;
        }
    }
}

template <int NT, int WC, int WP, int FC, int FP, int MODE>
static int launch_conv(ConvArgs &a, hipStream_t s) {
    const bool cat_only = a.adv && a.os == 1 && !a.oy && !a.ox && a.Ho == a.Hf && a.Wo == a.Wf;
    const bool plain = MODE == 2 && (!a.adv || cat_only) && !a.res_up;
    constexpr int CT = WC * FC * 32, PT = WP * FP * 32;
    ++g_launch_count;
    g_last_kernel = MODE == 1 ? MD_CONV_KERNEL_IGEMM_GENERIC_K : (CT == 128 && PT == 128 ? MD_CONV_KERNEL_IGEMM_128 :
                    (CT < 128 ? MD_CONV_KERNEL_IGEMM_SMALL_COUT : MD_CONV_KERNEL_OTHER));
    a.n_ctiles = (a.Cout + CT - 1) / CT;
    a.n_ptiles = (a.M + PT - 1) / PT;
    // one staging buffer is enough when the whole K fits one tile (1x1 convs on 64 channels): more
    // workgroups per CU for the HBM-bound layers
    if (MODE == 0) a.single_buf = 0;
    const int nbuf = a.Kpad / BK > 1 && !a.single_buf ? 2 : 1;
    const int tile_bytes = (CT + PT) * ROWB * nbuf;
    constexpr int ep_bytes = PT * (CT * 2 + 16);
    a.bias_lds_off = tile_bytes > ep_bytes ? tile_bytes : ep_bytes;
    const int lds = a.bias_lds_off + CT * 4;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    const long long blocks = (long long)a.n_ctiles * a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    constexpr bool HAS_PLAIN = MODE == 2;
    auto k = conv_igemm_kernel<NT, WC, WP, FC, FP, MODE, 1>;
    if (plain) k = a.relu == 2 ? conv_igemm_kernel<NT, WC, WP, FC, FP, MODE, HAS_PLAIN ? 2 : 1> : conv_igemm_kernel<NT, WC, WP, FC, FP, MODE, HAS_PLAIN ? 0 : 1>;
    if (lds > 64 * 1024 && ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NT), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}


static int launch_conv_dual_pingpong(ConvArgs &a, hipStream_t s);   // defined behind the ping-pong kernel

// 128 x 128 single-buffer kernel on the K-concatenation of two inputs (md_conv1x1_dual)
static int launch_conv_dual(ConvArgs &a, hipStream_t s, const Tune &tn) {
    // long-K, MFMA-bound forms (768 -> 1024, 1536 -> 2048 of the ResNet-50 stages 3 / 4): the 256x256 ping-pong kernel reading its K tiles
    // past nk_a from the second tensor
    if (a.Cout % 256 == 0 && a.Kpad >= tn.dual_pp_min_k && !a.res && (long long)(a.M + 255) / 256 * (a.Cout / 256) >= 256)
        return launch_conv_dual_pingpong(a, s);
    constexpr int CT = 128, PT = 128;
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_IGEMM_128;
    a.n_ctiles = (a.Cout + CT - 1) / CT;
    a.n_ptiles = (a.M + PT - 1) / PT;
    a.single_buf = 1;
    const int tile_bytes = (CT + PT) * ROWB;
    constexpr int ep_bytes = PT * (CT * 2 + 16);
    a.bias_lds_off = tile_bytes > ep_bytes ? tile_bytes : ep_bytes;
    const int lds = a.bias_lds_off + CT * 4;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    const long long blocks = (long long)a.n_ctiles * a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    auto k = conv_igemm_kernel<256, 2, 2, 2, 2, 2, 0, 1>;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

// ------------------------------------------------------------------------------------------------------------
// conv1x1_stream_kernel -- pointwise conv (1x1 / stride 1 / pad 0) with K = Cin <= 512, WEIGHT-STATIONARY.
//
// Why (r02, profiles/r02_conv_layers.json + DESIGN 6b): the 1x1 expand / reduce / lateral layers of the ResNet stages are HBM-bound
// (256->1024 + residual: 4.6 KB of activations per pixel against 0.5 MFLOP), yet the 128x128 kernel runs them at 3.9-5.1 TB/s: each
// of its workgroups re-stages its weight tile (as many bytes as the activation tile at K = 256) and re-reads the activation tile
// once per cout tile through the CU's vector-memory path, and its K loop is a chain of dependent round trips.  Here the weights of
// a (cout tile) never move after the prologue: each of the 4 waves keeps its CB x 32 cout rows x K as MFMA A fragments in REGISTERS
// (K = 256, CB = 2: 128 VGPRs), and the workgroup streams consecutive 32-pixel tiles past them:
//   * activations: one LDS-DMA stream into a ring of NR slots (32 px x K), requested NR - 1 tiles ahead, retired with counted
//     vmcnt + one raw s_barrier per tile; B fragments are conflict-free ds_read_b128 (16-B chunk index XOR row & 15, applied to the
//     DMA's source chunk and to the reads);
//   * residual: LDS-DMAed straight INTO the wave's private epilogue image (source chunks XOR-swizzled) one tile ahead; the epilogue
//     adds it in place in accumulator layout (each 8-byte cell is read and re-written by the same lane), so it costs no registers
//     and no second image;
//   * output: the image leaves as whole 16-B / 128-B-line non-temporal buffer stores.
// Every global access goes through a buffer descriptor re-based per tile (scalar 64-bit math only), so rows past M read zeros /
// drop their stores by the hardware range check and no tensor-size limit applies.
// Bytes through the CU's vector memory path per 32 px x 256 cout: x 16 K + residual 16 K + y 16 K = 48 KiB, against 96 KiB for the
// same outputs on the 128x128 kernel.  Rounds exactly where conv_igemm_kernel rounds (conv + bias -> bf16, + residual -> bf16).
// ------------------------------------------------------------------------------------------------------------
#define MD_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")

// RES 0: no residual; 1: residual with the output's layout (or a channel slice, a.Rs); 2: residual [N, ceil(Ho/2), ceil(Wo/2), Cout] read with
// nearest 2x upsampling (a.res_up: the FPN top-down add fused into the lateral conv)
// NW = waves per workgroup (cout tile = NW * CB * 32).  8 for K = 512 with Cout % 256 == 0 (r03): with 4 waves (CB = 1: 128 couts) every activation
// tile was staged by Cout / 128 workgroups, and PMC showed the second reader of a 256-cout layer missing L2 for half of it (FETCH 1.46 x the
// activation bytes, launch 1.35-1.53 x algorithmic); one 8-wave workgroup stages the tile once for 256 couts, same waves and LDS per CU.
template <int K, int CB, bool SILU, int RES, int NW = 4>
__global__ __launch_bounds__(NW * 64, 2) void conv1x1_stream_kernel(ConvArgs a, int tpw, int n_chunks, int chunks_per_xcd) {
    typedef __attribute__((address_space(3))) void lds_void;
    constexpr int KS = K / 16, RB = K * 2;          // MFMA k-steps, bytes per activation row
    constexpr int PT = 32, SLOT = PT * RB;          // pixels per tile, bytes per ring slot
    constexpr int NR = K == 512 ? (NW == 8 ? 3 : 2) : (K == 128 ? 4 : 3), D = NR - 1;   // ring slots, tiles of look-ahead (one 8-wave workgroup per CU: 3)
    constexpr int ND = SLOT / (NW * 1024);          // x DMA instructions per wave and tile (1 KiB each)
    constexpr int RPI = RB >= 1024 ? 1 : 1024 / RB; // tile rows per DMA instruction
    constexpr int CW = CB * 32;                     // couts per wave (4 waves: CT = 4 * CW)
    constexpr int CPP = CB * 4, EROW = CPP * 16;    // 16-B chunks / bytes per row of the wave's epilogue image [32 px][CW]
    constexpr int EW = PT * EROW, NE = EW / 1024;   // image bytes; 1-KiB pieces = residual DMAs = store instructions per tile
    constexpr int RPE = 1024 / EROW;                // image rows per piece
    constexpr int ESH = CPP == 8 ? 1 : 2;           // image swizzle: chunk ^= (row >> ESH) & (CPP - 1)
    constexpr int NRES = RES != 0 ? NE : 0;
    static_assert(K % 128 == 0 && K <= 512 && CB * K <= 512, "weights must fit 128 registers per lane");
    static_assert(ND >= 1 && ND * NW * 1024 == SLOT, "the x tile must split into whole 1-KiB pieces per wave");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ring = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    char *E = smem + NR * SLOT + wave * EW;
    float *bias_lds = reinterpret_cast<float *>(smem + NR * SLOT + NW * EW);

    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int ct = slot_id % a.n_ctiles, chunk = xcd * chunks_per_xcd + slot_id / a.n_ctiles;
    if (chunk >= n_chunks) return;
    const int t0 = chunk * tpw;
    const int nt = a.n_ptiles - t0 < tpw ? a.n_ptiles - t0 : tpw;
    if (nt <= 0) return;
    const int cout_w = ct * (NW * CW) + wave * CW;  // this wave's first output channel
    const int r_stride = a.Rs ? a.Rs : a.Ctot;      // residual pixel stride (channels)
    const int r_c0 = a.Rs ? cout_w : a.c_off + cout_w;
    const int y_c0 = a.c_off + cout_w;

    // ---- per-lane offsets, constant for the whole launch
    unsigned xoff[ND];      // x DMA i of this wave: tile row, source chunk (swizzled)
    int xdst[ND];
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const int j = wave * ND + i, byte = lane * 16;
        const int row = j * RPI + (RB >= 1024 ? 0 : byte / RB), pc = (byte % RB) >> 4;
        xoff[i] = (unsigned)(row * a.Xs * 2 + ((pc ^ (row & 15)) << 4));
        xdst[i] = j * 1024;
    }
    unsigned yoff[NE], roff[NE];   // image piece i: row = i * RPE + lane / CPP, physical chunk lane % CPP holds logical chunk ^ swizzle
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int row = i * RPE + lane / CPP, lc = (lane % CPP) ^ ((row >> ESH) & (CPP - 1));
        yoff[i] = (unsigned)(row * a.Ctot * 2 + lc * 16);
        roff[i] = (unsigned)(row * r_stride * 2 + lc * 16);
    }
    int foff[8];            // B fragment of k-step s: ring slot + foff[s & 7] + (s >> 3) * 256
#pragma unroll
    for (int j = 0; j < 8; ++j) foff[j] = lr * RB + (((2 * j + lh) ^ (lr & 15)) << 4);
    const int eswz = (lr >> ESH) & (CPP - 1);

    // ---- per-tile descriptors (scalar).  A tile index past this workgroup's range gets an EMPTY descriptor: its DMAs are still
    // issued (zero fill of a slot nobody reads), so every wave's vector-memory queue has the same shape in every iteration and
    // the waits below are compile-time counts.
    auto clip = [](long long rem) { return (int)(rem > 0x7fffffffLL ? 0x7fffffffLL : (rem < 0 ? 0 : rem)); };
    auto x_desc = [&](int t) {
        const long long b = (long long)(t0 + t) * PT * a.Xs * 2;
        const long long rem = t < nt ? (long long)a.x_bytes - b : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void *)((const char *)a.x + (t < nt ? b : 0)), 0, clip(rem), 0x00020000);
    };
    auto r_desc = [&](int t) {
        const long long m0 = t < nt ? (long long)(t0 + t) * PT : 0;
        const long long rem = t < nt ? ((long long)a.M - m0) * r_stride * 2 - r_c0 * 2 : 0;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(a.res + m0 * r_stride + r_c0), 0, clip(rem), 0x00020000);
    };
    auto y_desc = [&](int t) {
        const long long m0 = (long long)(t0 + t) * PT;
        const long long rem = ((long long)a.M - m0) * a.Ctot * 2 - y_c0 * 2;
        return __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + m0 * a.Ctot + y_c0), 0, clip(rem), 0x00020000);
    };
    auto dma_x = [&](int t, int slot) {
        __amdgpu_buffer_rsrc_t rs = x_desc(t);
        char *dst = ring + slot * SLOT;
        if (a.tune & 1) {
#pragma unroll
            for (int i = 0; i < ND; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(dst + xdst[i]), 16, (int)xoff[i], 0, 0, 2);
        } else {
#pragma unroll
            for (int i = 0; i < ND; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(dst + xdst[i]), 16, (int)xoff[i], 0, 0, 0);
        }
    };
    // RES 2: per-lane (n, ho, wo) of the lane's NE image rows, advanced by one tile (32 pixels) per request
    int u_wo[RES == 2 ? NE : 1], u_ho[RES == 2 ? NE : 1], u_n[RES == 2 ? NE : 1], u_m[RES == 2 ? NE : 1];
    const int Hr = (a.Ho + 1) >> 1, Wr = (a.Wo + 1) >> 1;
    if constexpr (RES == 2) {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int m = t0 * PT + i * RPE + lane / CPP;
            const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
            u_m[i] = m; u_n[i] = n; u_ho[i] = r / a.Wo; u_wo[i] = r - u_ho[i] * a.Wo;
        }
    }
    auto dma_res_up = [&](bool live) {   // requests the residual rows of the NEXT tile in sequence (tiles are consecutive)
        const long long tot = (long long)a.N * Hr * Wr * a.Cout * 2;
        __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)a.res, 0, live ? clip(tot) : 0, 0x00020000);
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int lc = (lane % CPP) ^ (((i * RPE + lane / CPP) >> ESH) & (CPP - 1));
            const unsigned off = u_m[i] < a.M ? (unsigned)((((u_n[i] * Hr + (u_ho[i] >> 1)) * Wr + (u_wo[i] >> 1)) * a.Cout + cout_w) * 2 + lc * 16)
                                              : 0x80000000u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(E + i * 1024), 16, (int)off, 0, 0, 0);
            u_m[i] += PT; u_wo[i] += PT;
            while (u_wo[i] >= a.Wo) { u_wo[i] -= a.Wo; if (++u_ho[i] == a.Ho) { u_ho[i] = 0; ++u_n[i]; } }
        }
    };
    auto dma_res = [&](int t) {
        if constexpr (RES == 2) { dma_res_up(t < nt); return; }
        __amdgpu_buffer_rsrc_t rs = r_desc(t);
        if (a.tune & 2) {
#pragma unroll
            for (int i = 0; i < NE; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(E + i * 1024), 16, (int)roff[i], 0, 0, 2);
        } else {
#pragma unroll
            for (int i = 0; i < NE; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void *)(E + i * 1024), 16, (int)roff[i], 0, 0, 0);
        }
    };

    // ---- prologue: the first D activation tiles and the first residual tile are requested BEFORE the weights
#pragma unroll
    for (int t = 0; t < D; ++t) dma_x(t, t);
    if constexpr (RES != 0) dma_res(0);
    if (tid < NW * CW) bias_lds[tid] = a.bias[ct * (NW * CW) + tid];
    bf16x8 wr[CB][KS];
#pragma unroll
    for (int b = 0; b < CB; ++b)
#pragma unroll
        for (int s = 0; s < KS; ++s)
            wr[b][s] = *reinterpret_cast<const bf16x8 *>(a.w + (size_t)(cout_w + b * 32 + lr) * a.Kpad + s * 16 + lh * 8);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int s_cur = 0, s_new = D;    // ring slot of tile t / of tile t + D
    for (int t = 0; t < nt; ++t) {
        // x tile t was requested D iterations ago; behind it in this wave's queue: per iteration NE stores and NRES residual DMAs,
        // and (D - 1 times) the ND pieces of a later x tile
        __builtin_amdgcn_sched_barrier(0);
        MD_WAIT_VMCNT(D * (NE + NRES) + (D - 1) * ND);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        dma_x(t + D, s_new);                // that slot was last read in iteration t - 1: every wave is past it (barrier)

        const char *S = ring + s_cur * SLOT;
        f32x16 acc[CB];
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        bf16x8 fb[2];
        fb[0] = *reinterpret_cast<const bf16x8 *>(S + foff[0]);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + 1 < KS) fb[(s + 1) & 1] = *reinterpret_cast<const bf16x8 *>(S + foff[(s + 1) & 7] + ((s + 1) >> 3) * 256);
#pragma unroll
            for (int b = 0; b < CB; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wr[b][s], fb[s & 1], acc[b], 0, 0, 0);
        }
        s_cur = s_cur + 1 == NR ? 0 : s_cur + 1;
        s_new = s_new + 1 == NR ? 0 : s_new + 1;

        // ---- epilogue on the wave's private image: residual (already there, behind it only this iteration's x pieces) +
        // bf16(acc + bias), in place.  Every LDS access below uses a builtin vector type: hipcc's waitcnt pass puts an
        // s_waitcnt vmcnt(0) in front of an LDS access WITHOUT type-based alias info while LDS-DMAs are pending (seen with
        // float4 / uint2: it drained the x look-ahead every tile); the counted waits here are the synchronisation.
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (RES != 0) MD_WAIT_VMCNT(ND);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < CB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + wave * CW + b * 32 + 8 * g + 4 * lh);
                f32x2 s01 = (f32x2){acc[b][4 * g + 0], acc[b][4 * g + 1]} + (f32x2){bv.x, bv.y};
                f32x2 s23 = (f32x2){acc[b][4 * g + 2], acc[b][4 * g + 3]} + (f32x2){bv.z, bv.w};
                if (SILU) { s01.x = silu(s01.x); s01.y = silu(s01.y); s23.x = silu(s23.x); s23.y = silu(s23.y); }
                u32x2 pk;
                pk.x = pk_bf16(s01.x, s01.y);
                pk.y = pk_bf16(s23.x, s23.y);
                u32x2 *cell = reinterpret_cast<u32x2 *>(E + lr * EROW + (((4 * b + g) ^ eswz) << 4) + 8 * lh);
                if constexpr (RES != 0) {
                    const u32x2 rv = *cell;
                    const f32x2 a01 = (f32x2){__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xffff0000u)} +
                                      (f32x2){__uint_as_float(rv.x << 16), __uint_as_float(rv.x & 0xffff0000u)};
                    const f32x2 a23 = (f32x2){__uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xffff0000u)} +
                                      (f32x2){__uint_as_float(rv.y << 16), __uint_as_float(rv.y & 0xffff0000u)};
                    pk.x = pk_bf16(a01.x, a01.y);
                    pk.y = pk_bf16(a23.x, a23.y);
                }
                if (a.relu == 1) { pk.x = pk_relu_bf16(pk.x); pk.y = pk_relu_bf16(pk.y); }
                *cell = pk;
            }
        __amdgpu_buffer_rsrc_t ry = y_desc(t);
        MD_WAVE_LDS_ORDER();   // the cells above were written by other lanes of this wave than the ones that read them out below
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            u32x4 v = *reinterpret_cast<const u32x4 *>(E + i * 1024 + lane * 16);
            if (a.tune & 4) MD_BUFFER_STORE_B128(v, ry, yoff[i], 0, 2);
            else MD_BUFFER_STORE_B128(v, ry, yoff[i], 0, 0);
        }
        // the image is free once its read-out has reached the registers: request the next tile's residual into it
        if constexpr (RES != 0) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_res(t + 1);
        }
    }
}

// dispatch + launch of conv1x1_stream_kernel; MD_ERR_UNSUPPORTED_STREAM when the layer is not one it takes
#define MD_ERR_UNSUPPORTED_STREAM 101
template <int K, int CB, int NW = 4>
static int launch_conv1x1_stream_t(ConvArgs &a, hipStream_t s, const Tune &tn) {
    constexpr int NR = K == 512 ? (NW == 8 ? 3 : 2) : (K == 128 ? 4 : 3);
    constexpr int CT = NW * CB * 32;
    const int lds = NR * 32 * K * 2 + NW * (32 * CB * 64) + CT * 4;
    a.n_ctiles = a.Cout / CT;
    a.n_ptiles = (a.M + 31) / 32;
    const long long slots = 256LL * tn.stream_wgs_per_cu * 4 / NW * tn.stream_rounds;   // resident workgroups (two 4-wave ones per CU) x rounds
    a.tune = tn.stream_cache_bits;
    long long tpw = ((long long)a.n_ptiles * a.n_ctiles + slots - 1) / slots;
    if (tpw < 4) tpw = 4;
    const long long n_chunks = (a.n_ptiles + tpw - 1) / tpw;
    const long long chunks_per_xcd = (n_chunks + 7) / 8;
    const long long blocks = chunks_per_xcd * 8 * a.n_ctiles;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    auto k = a.res ? (a.relu == 2 ? conv1x1_stream_kernel<K, CB, true, 1, NW> : conv1x1_stream_kernel<K, CB, false, 1, NW>)
                   : (a.relu == 2 ? conv1x1_stream_kernel<K, CB, true, 0, NW> : conv1x1_stream_kernel<K, CB, false, 0, NW>);
    if (a.res_up) k = conv1x1_stream_kernel<K, CB, false, 2, NW>;
    if (ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_STREAM_1X1;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(NW * 64), lds, s, a, (int)tpw, (int)n_chunks, (int)chunks_per_xcd);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

static bool stream1x1_takes(const ConvArgs &a) {
    const bool cat_only = !a.adv || (a.os == 1 && a.oy == 0 && a.ox == 0 && a.Ho == a.Hf && a.Wo == a.Wf);
    if (!a.pointwise || !cat_only || a.Kpad != a.Cin || a.x_bytes == 0) return false;
    // upsampled residual (the FPN lateral convs): plain output, no SiLU, residual tensor inside the 32-bit offset reach
    if (a.res_up && (a.adv || a.relu == 2 || (long long)a.N * ((a.Ho + 1) / 2) * ((a.Wo + 1) / 2) * a.Cout * 2 >= 0x7fff0000LL)) return false;
    if (a.Cin != 128 && a.Cin != 256 && a.Cin != 512) return false;
    return a.Cout % 128 == 0;
}

static int launch_conv1x1_stream(ConvArgs &a, hipStream_t s, const Tune &tn) {
    if (!stream1x1_takes(a)) return MD_ERR_UNSUPPORTED_STREAM;
    const bool wide = a.Cout % 256 == 0;
    if (a.Cin == 128) return wide ? launch_conv1x1_stream_t<128, 2>(a, s, tn) : launch_conv1x1_stream_t<128, 1>(a, s, tn);
    if (a.Cin == 256) return wide ? launch_conv1x1_stream_t<256, 2>(a, s, tn) : launch_conv1x1_stream_t<256, 1>(a, s, tn);
    // K = 512: 256 couts per (8-wave) workgroup where Cout allows, so that an activation tile is staged once per 256 couts
    if (wide && !tn.stream_narrow) return launch_conv1x1_stream_t<512, 1, 8>(a, s, tn);
    return launch_conv1x1_stream_t<512, 1>(a, s, tn);
}

// ------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with HALO REUSE.
//
// Measured on the generic kernel (tools/conv_ab.py ablations + tools/ubench/dma_bench.hip, r01): the 128x128
// implicit-GEMM tile is bound by L2->LDS staging (LDS-DMA tops out near 110-120 GB/s per CU), not by MFMA, and a
// 3x3 conv stages every activation 9 times (once per tap).  Here a workgroup owns an 8x16 block of output pixels
// of one image: per 64-channel chunk it stages the 10x18 input halo ONCE and runs all 9 taps out of it with
// shifted fragment reads; only the 16 KiB weight slice of each (chunk, tap) step is streamed.  Staged bytes per
// step drop from 32 KiB to 16 KiB + 23 KiB/9 = 18.6 KiB.  Weights must be packed with korder 1
// ([Cout][ci/64][tap][64]) so that a step's slice is contiguous.
// ------------------------------------------------------------------------------------------------------------
constexpr int HT_H = 8, HT_W = 16, HALO_W = HT_W + 2, HALO_ROWS = (HT_H + 2) * HALO_W;  // 180 halo pixels
constexpr int HALO_DMAS = (HALO_ROWS + 7) / 8;                                           // 23 x 1 KiB
constexpr int HALO_BYTES = HALO_DMAS * 1024;

// CT = cout tile (128: waves 2 x 2, each 64 couts x 64 pixels; 64: waves 1 x 4, each 64 couts x 32 pixels).
// ONE_HALO: a single halo buffer, refilled (exposed, but four workgroups share the CU) when the 64-channel chunk
// changes: 2 x 8 KiB weight stages + 23 KiB halo = 39 KiB -> four workgroups per CU for CT 64.
template <int CT, bool ONE_HALO>
__global__ __launch_bounds__(256, CT == 64 ? 4 : 2) void conv3x3_halo_kernel(ConvArgs a, int tiles_x, int tiles_y) {
    constexpr int PT = HT_H * HT_W, WC = CT / 64, WP = 4 / WC, FC = 2, FP = 4 / WP;
    static_assert(WC * WP == 4 && WC * FC * 32 == CT && WP * FP * 32 == PT, "wave grid must cover the tile");
    constexpr int WSTAGE = CT * ROWB;
    constexpr int EP_STRIDE = CT * 2 + 16;
    constexpr unsigned OOR = 0x80000000u;
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *Wst = smem;                // 2 weight stages
    char *Hst = smem + 2 * WSTAGE;   // 1 or 2 halo stages

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave / WP, wp = wave % WP;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = slot % a.n_ctiles, pt = xcd * a.pt_per_xcd + slot / a.n_ctiles;
    if (pt >= a.n_ptiles) return;
    const int cout0 = ct * CT;
    const int tx = pt % tiles_x, ty = (pt / tiles_x) % tiles_y, n = pt / (tiles_x * tiles_y);
    const int y0 = ty * HT_H, x0 = tx * HT_W;
    const float bias_early = tid < CT ? a.bias[cout0 + tid] : 0.f;  // parked in LDS for the epilogue (see conv_igemm_kernel)

    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);

    // weight staging: thread -> rows row0 + 32 i, logical chunk `chunk` (as in conv_igemm_kernel)
    const int row0 = tid >> 3;
    const int chunk = (tid & 7) ^ ((row0 >> 1) & 7);
    const int a_off0 = ((cout0 + row0) * a.Kpad + chunk * 8) * 2;
    const int wrow = wave * 8;
    // halo staging: this wave issues DMA q = wave + 4 j (j < 6); lane -> halo row 8 q + lane/8
    unsigned h_off[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int q = wave + 4 * j, r = q * 8 + (lane >> 3);
        const int hy = r / HALO_W, hx = r - hy * HALO_W;
        const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
        const int lchunk = (lane & 7) ^ ((r >> 1) & 7);
        const bool ok = q < HALO_DMAS && r < HALO_ROWS && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        h_off[j] = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Xs + lchunk * 8) * 2) : OOR;
    }
    const int n_chunks = a.Cin / 64;
    auto dma_weights = [&](int step, int buf) {
#pragma unroll
        for (int i = 0; i < CT / 32; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(Wst + buf * WSTAGE + (wrow + 32 * i) * ROWB), 16,
                                                     a_off0 + i * (32 * a.Kpad * 2), step * (BK * 2), 0, 0);
    };
    auto dma_halo_piece = [&](int c, int j, int buf) {  // piece j (0..5) of chunk c's halo
        const int q = wave + 4 * j;
        if (q < HALO_DMAS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(Hst + buf * HALO_BYTES + q * 1024), 16, (int)h_off[j],
                                                     c * 128, 0, 0);
    };

    f32x16 acc[FC][FP];
#pragma unroll
    for (int i = 0; i < FC; ++i)
#pragma unroll
        for (int j = 0; j < FP; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int lr = lane & 31, lh = lane >> 5;
    int fa_off[BK / 16];
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) fa_off[kk] = swz(wc * 64 + lr, kk * 2 + lh);
    // lane -> pixel of a 32-pixel fragment (= two 16-pixel tile rows, 18 halo rows apart): ds_read_b128 is served in the lane groups
    // {0-3, 12-15, 20-27} {4-11, 16-19, 28-31} (+32), 16 lanes x 16 B = all 64 banks per pass.  With pixel = lane a group straddles the two
    // tile rows and two of its halo rows coincide mod 16 (a 2-way conflict on every tap: SQ_LDS_BANK_CONFLICT 1.3 x the LDS-busy cycles,
    // tools/pmc_lds_survey.sh); this permutation gives each group one whole tile row = 16 consecutive halo rows
    const int hp = (int)(((0x73261540u >> ((lr >> 2) * 4)) & 7u) << 2) | (lr & 3);   // 4-lane blocks 0..7 -> 0, 4, 5, 1, 6, 2, 3, 7
    // halo row of this lane's pixel for tap (0,0), per pixel fragment j
    int r0[FP];
#pragma unroll
    for (int j = 0; j < FP; ++j) {
        const int p = (wp * FP + j) * 32 + hp;
        r0[j] = (p >> 4) * HALO_W + (p & 15);
    }

    // prologue: halo of chunk 0 (all six pieces) + weights of step 0
#pragma unroll
    for (int j = 0; j < 6; ++j) dma_halo_piece(0, j, 0);
    dma_weights(0, 0);
    const int n_steps = n_chunks * 9;
    int c = 0, t = 0;  // chunk / tap of the step being computed
    for (int s_ = 0; s_ < n_steps; ++s_) {
        if (ONE_HALO && t == 0 && c > 0) {
            __syncthreads();  // every wave has finished the last tap of chunk c - 1: the halo buffer may be refilled
#pragma unroll
            for (int j = 0; j < 6; ++j) dma_halo_piece(c, j, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (s_ + 1 < n_steps) dma_weights(s_ + 1, (s_ + 1) & 1);
        if (!ONE_HALO && t < 6 && c + 1 < n_chunks) dma_halo_piece(c + 1, t, (c + 1) & 1);
        // compute step (c, t): A from the weight stage, B from the halo stage at the tap's shifted rows
        const char *Wb = Wst + (s_ & 1) * WSTAGE;
        const char *Hb = Hst + (ONE_HALO ? 0 : (c & 1) * HALO_BYTES);
        const int radd = (t / 3) * HALO_W + (t % 3);
        int b_off[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j) {
            const int r = r0[j] + radd, sw = (r >> 1) & 7;
            b_off[j] = r * ROWB + (((sw & 6) | (lh ^ (sw & 1))) << 4);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            bf16x8 fa[FC], fb[FP];
#pragma unroll
            for (int i = 0; i < FC; ++i) fa[i] = *reinterpret_cast<const bf16x8 *>(Wb + fa_off[kk] + i * 32 * ROWB);
#pragma unroll
            for (int j = 0; j < FP; ++j) fb[j] = *reinterpret_cast<const bf16x8 *>(Hb + (b_off[j] ^ (kk << 5)));
#pragma unroll
            for (int i = 0; i < FC; ++i)
#pragma unroll
                for (int j = 0; j < FP; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (++t == 9) { t = 0; ++c; }
    }
    __syncthreads();

    // ---- epilogue (same scheme as conv_igemm_kernel): bias -> bf16x4 -> LDS [pixel][cout] -> 16-B NHWC stores
    constexpr int CPP = CT / 8, EP_ITERS = PT * CPP / 256;
    auto pix_off = [&](int p_local, int cglob, int cstride) -> long long {
        const int y = y0 + (p_local >> 4), x = x0 + (p_local & 15);
        if (y >= a.H || x >= a.W || cglob >= a.Cout) return -1;
        return ((long long)(n * a.H + y) * a.W + x) * cstride + cglob;
    };
    u32x4 rres[EP_ITERS];
    if (a.res) {
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int e = tid + it * 256;
            const long long off = pix_off(e / CPP, cout0 + (e % CPP) * 8, a.Rs ? a.Rs : a.Cout);
            rres[it] = (u32x4){0u, 0u, 0u, 0u};
            if (off >= 0) rres[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.res + off));
        }
    }
    char *E = smem;
    float *bias_lds = reinterpret_cast<float *>(smem + PT * EP_STRIDE);
    if (tid < CT) bias_lds[tid] = bias_early;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FC; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c_local = (wc * FC + i) * 32 + 8 * g + 4 * lh;
            const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + c_local);
#pragma unroll
            for (int j = 0; j < FP; ++j) {
                const int p_local = (wp * FP + j) * 32 + hp;
                f32x2 s01 = (f32x2){acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]} + (f32x2){bv.x, bv.y};
                f32x2 s23 = (f32x2){acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} + (f32x2){bv.z, bv.w};
                if (a.relu == 2) { s01.x = silu(s01.x); s01.y = silu(s01.y); s23.x = silu(s23.x); s23.y = silu(s23.y); }
                uint2 pk;
                pk.x = pk_bf16(s01.x, s01.y);
                pk.y = pk_bf16(s23.x, s23.y);
                if (a.relu == 1 && !a.res) { pk.x = pk_relu_bf16(pk.x); pk.y = pk_relu_bf16(pk.y); }
                *reinterpret_cast<uint2 *>(E + p_local * EP_STRIDE + c_local * 2) = pk;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < EP_ITERS; ++it) {
        const int e = tid + it * 256;
        const int p_local = e / CPP, cc = e % CPP;
        long long off = pix_off(p_local, cout0 + cc * 8, a.Ctot);   // concat output: channels [c_off, c_off + Cout) of Ctot
        if (off < 0) continue;
        off += a.c_off;
        u32x4 v = *reinterpret_cast<const u32x4 *>(E + p_local * EP_STRIDE + cc * 16);
        if (a.res) {
            const u32x4 rv = rres[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 sum = (f32x2){__uint_as_float(v[k] << 16), __uint_as_float(v[k] & 0xffff0000u)} +
                                  (f32x2){__uint_as_float(rv[k] << 16), __uint_as_float(rv[k] & 0xffff0000u)};
                v[k] = pk_bf16(sum.x, sum.y);
                if (a.relu == 1) v[k] = pk_relu_bf16(v[k]);
            }
        }
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(a.y + off));
    }
}

template <int CT, bool ONE_HALO>
static int launch_conv3x3_halo(ConvArgs &a, hipStream_t s) {
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_HALO;
    const int tiles_x = (a.W + HT_W - 1) / HT_W, tiles_y = (a.H + HT_H - 1) / HT_H;
    a.n_ctiles = (a.Cout + CT - 1) / CT;
    a.n_ptiles = a.N * tiles_x * tiles_y;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    const long long blocks = (long long)a.n_ctiles * a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    const int stage = 2 * CT * ROWB + (ONE_HALO ? 1 : 2) * HALO_BYTES;
    const int ep = HT_H * HT_W * (CT * 2 + 16) + CT * 4;  // epilogue image + bias copy
    const int lds = stage > ep ? stage : ep;
    auto k = conv3x3_halo_kernel<CT, ONE_HALO>;
    if (ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, s, a, tiles_x, tiles_y);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}


// ------------------------------------------------------------------------------------------------------------
// 256(cout) x 256(pixel) x 64 PING-PONG kernel for the MFMA-bound layers (Cin % 64 == 0, Cout % 256 == 0).
//
// Why: on the 128x128 kernels a K tile's LDS-DMA batch and its MFMAs do not overlap (r01 ablations) and one
// barrier per tile drains the DMA queue (vmcnt 0).  Here ONE workgroup of 8 waves owns a CU.  Waves 0-3 (cout rows
// 0-127) and waves 4-7 (rows 128-255) sit pairwise on the four SIMDs and run HALF A PHASE APART (group 1 passes one
// extra barrier up front, group 0 one at the end): while one group runs its 8-MFMA cluster at raised priority the
// other issues its fragment reads and its share of the LDS-DMA.  A K tile is two phases (16 MFMAs = one 64-row half of the
// wave's 128 x 64 output each) and four half tiles {A0, B0, B1, A1} x two parities = 128 KiB of LDS; the DMA stream
// runs with COUNTED waits (vmcnt 8 = four half tiles stay in flight across every barrier).
//
// Hazard bookkeeping (p = phase index, groups staggered by one barrier; a phase's fragment reads are RETIRED
// (lgkmcnt 0) before its first barrier):
//   RAW  a vmcnt wait placed before the first barrier of phase p covers reads issued in phase p+1 by BOTH groups;
//   WAR  a buffer may be re-staged in phase p when its last ds_read was issued in phase <= p-1.
//   tile t:  ph0 reads A0,B0,B1 / stages (t+1,A1)      ph1 reads A1 / stages (t+2,A0), (t+2,B0), (t+2,B1)
//   every DMA is waited for two phases after its issue: vmcnt(8) = the four youngest half tiles stay in flight.
// Half tile "A0" = cout rows {0-63, 128-191} (the first 64 rows of each wave group), "A1" the others; "B0" = the
// first 32 pixels of each of the four 64-pixel wave columns, "B1" the others -- so every wave reads every half tile.
// ------------------------------------------------------------------------------------------------------------
constexpr int PP_HALF = 128 * ROWB;  // 16 KiB
struct KWalk { int tap, kh, kw, cc0; };

// MF 0: v_mfma_f32_32x32x16_bf16, MF 1: v_mfma_f32_16x16x32_bf16 (same LDS image, reads and cycles per flop; the chip holds a
// different clock on the two shapes under load -- MI355X_MICROARCH.md DVFS item 7 -- so both are built and the faster kept).
// PERS (GEN 0 / 2, no residual, plain or concat output): PERSISTENT form -- one workgroup per CU walks over several pixel tiles; the seven
// prologue half tiles of the NEXT tile are requested before the current tile's epilogue, which runs barrier-free through wave-private
// 2.5-KiB LDS slabs (the staging buffers stay free for the DMA stream) and leaves as buffer stores.
// HALO (3x3 / stride 1 / pad 1, korder-1 weights, GEN 0 / 2; r03): the pixel tile is a 16 x 16 block of ONE image (an 8 x 32 block in the
// bottom strip of an image whose height leaves at most 8 rows past the 16-row tiles: 200 x 336 is covered with 0.2 % idle pixels) and its
// 18 x 18 (10 x 34) x 64-channel halo is staged ONCE per channel chunk (41 / 43 pieces of 1 KiB, two buffers alternating by chunk parity); the
// nine taps of the chunk = nine K tiles read their B fragments out of it with shifted rows.  The B stream drops from 32 KiB to 4.6 KiB per K
// tile (8 -> 5 LDS-DMA pieces per wave and K tile, the fifth being a halo piece or -- K tiles 6-8 of a chunk, pieces past the halo -- a 1-KiB
// zero fill of a dummy area, so that every K tile issues the same number and the counted waits stay immediates: vmcnt(5)).  Same K order, same
// MFMA sequence per output element as the linear-tile form: bit-identical results.
// LDS (HALO): A halves [0, 64K) = {parity} x {A0, A1}; halo buffers [64K, 107K), [107K, 150K); dummy 1K; bias 1K; W2 8K (HEAD) = 160 KiB.
constexpr int HB_ROWS = 344 /* 18 x 18 = 324 or 10 x 34 = 340 halo pixels, in whole pieces (8 rows) */, HB_BYTES = HB_ROWS * ROWB, HB0_OFF = 4 * 128 * ROWB, HB_DUMMY = HB0_OFF + 2 * HB_BYTES, HB_BIAS = HB_DUMMY + 1024,
              HB_W2 = HB_BIAS + 256 * 4, HB_LDS = HB_W2, HB_LDS_HEAD = HB_W2 + 16 * 256 * 2;
template <int ABL, int MF = 0, int GEN = 1, bool HEAD = false, bool PERS = false, bool HALO = false>  // GEN as in conv_igemm_kernel; HEAD: fused 1x1 head; ABL 0: product; timing ablations (wrong results): 1 no in-loop staging, 2 no output stores, 4 stamps
__global__ __launch_bounds__(512, 2) void conv_pingpong_kernel(ConvArgs a) {
    static_assert(!PERS || (!HEAD && ABL == 0 && GEN != 1), "the persistent form has the plain epilogues only");
    static_assert(!HALO || (ABL == 0 && GEN != 1), "the halo form has the plain epilogues only");
    static_assert(!(HALO && PERS && MF == 0), "persistent + HALO exists on the 16x16x32 shape only (registers; its slab epilogue has no lane permutation)");
    constexpr int CT = 256, PT = 256, NT = 512;
    constexpr int EP_STRIDE = CT * 2 + 16;
    constexpr unsigned OOR = 0x80000000u;
    constexpr int H_A0 = 0, H_B0 = 1, H_B1 = 2, H_A1 = HALO ? 1 : 3;
    constexpr int A_SLOTS = HALO ? 2 : 4;   // half-tile slots per K-tile parity
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int lr = lane & 31, lh = lane >> 5;
    const int l16 = lane & 15, lq = lane >> 4;
    // HALO: which pixel of a fragment a lane's accumulator column is.  The halo rows a fragment read touches start at ANY row (tap shifts), and
    // ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ... (MICROARCH, LDS): with lane = pixel the two k-chunks a
    // group mixes collide 2-way on 3 of 4 alignments (r03 PMC: SQ_LDS_BANK_CONFLICT 7 x the linear form's).  16x16x32: lanes 0-3 / 12-15 take the
    // EVEN pixels of the 16, lanes 4-11 the odd ones -- a group then reads one chunk on even rows and the other on odd rows, which live in
    // different halves of the 256-B bank window; 32x32x16: each group takes 16 consecutive pixels.  Conflict-free for every alignment
    // (brute-forced over all of them); the output is unchanged -- only which lane computes which pixel.
    const int hp16 = !HALO ? l16 : (l16 < 4 ? 2 * l16 : (l16 < 12 ? 2 * (l16 - 4) + 1 : 2 * (l16 - 8)));
    const int hp32 = !HALO ? lr : (lr < 4 ? lr : (lr < 12 ? lr + 12 : (lr < 16 ? lr - 8 : (lr < 20 ? lr + 8 : (lr < 28 ? lr - 12 : lr)))));
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int ct = slot % a.n_ctiles;
    int ptl = slot / a.n_ctiles;                                        // pixel tile inside the XCD's range
    const int pstep = PERS ? (int)(gridDim.x >> 3) / a.n_ctiles : 0;    // PERS: the workgroups of an (XCD, cout tile) stride through the range
    if (ptl >= a.pt_per_xcd || xcd * a.pt_per_xcd + ptl >= a.n_ptiles) return;
    const int cout0 = ct * CT;
    int pix0 = (xcd * a.pt_per_xcd + ptl) * PT;
    const int n_taps = a.kh * a.kw, nk = a.Kpad / BK;
    unsigned long long clk_start = 0;
    if (ABL == 4) clk_start = __builtin_readcyclecounter();
    // bias: requested now, parked in LDS past the epilogue image after the prologue wait (see conv_igemm_kernel)
    const float bias_early = tid < CT ? a.bias[cout0 + tid] : 0.f;
    float *bias_lds = reinterpret_cast<float *>(smem + (HALO ? HB_BIAS : (PERS ? 8 * 128 * ROWB : PT * EP_STRIDE)));   // PERS: right behind the 128 KiB of staging buffers

    __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)a.w, 0, a.w_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)a.x, 0, a.x_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void *)(a.x2 ? a.x2 : a.x), 0, a.x2 ? a.x2_bytes : 0u, 0x00020000);

    // ---- staging map: one wave instruction = 8 rows x 128 B; a half tile = 2 instructions per wave (rows srow, srow+64)
    const int srow = wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((srow >> 1) & 7);  // source-side swizzle (same for srow and srow + 64)
    const int a_off = ((cout0 + srow) * a.Kpad + chunk * 8) * 2;
    const int a_half = 64 * a.Kpad * 2, a_pass = 128 * a.Kpad * 2;
    int p_base[4];       // [hB * 2 + i]: byte offset of (n, hi0, wi0, chunk) of the staged pixel row
    unsigned p_taps[4];  // tap-validity bits (0 past M)
    unsigned p_base2[4]; // DUAL (a.x2 != null: md_conv1x1_dual): the row's pixel in the second tensor, sampled with a.stride2 (OOR past M)
    // HALO: image / top-left pixel / shape of the tile (t_twl = log2 of its width: 4 = 16 x 16, 5 = 8 x 32 of the bottom strip; t_hw = halo
    // width, t_magic: r / t_hw == (r * t_magic) >> 16 for r < 400), and this lane's byte offset into x for each of the wave's six halo pieces
    // (piece wave + 8 j = halo rows 8 * piece .. + 7, row r = halo pixel (r / t_hw, r % t_hw); OOR outside the image = the conv's zero padding)
    int t_n = 0, t_y0 = 0, t_x0 = 0, t_twl = 4, t_hw = 18, t_magic = 3641, t_hpix = 324;
    int hpb[4] = {0, 0, 0, 0};   // halo row of this lane's pixel for tap (0, 0): MF 1 one per 16-pixel fragment of the wave's four, MF 0 one per pixel half
    // (computed where a piece is issued, ~18 VALU per K tile: six precomputed offsets selected by the wave-uniform piece index ended up in
    // SCRATCH -- hipcc lowers the select chain to an indexed private-memory load, a vector-memory operation inside the K loop that
    // would also count in vmcnt)
    const int nch = a.Cin >> 6;
    auto piece_off = [&](int j, int c) -> unsigned {
        const int r = (wave + 8 * j) * 8 + (lane >> 3);
        const int hy = (r * t_magic) >> 16, hx = r - hy * t_hw;
        const int iy = t_y0 - 1 + hy, ix = t_x0 - 1 + hx;
        // the 16-B chunk swizzle is taken on the row's ABSOLUTE LDS row (the fragment reads fold the buffer base into the row index): buffer
        // 1 starts HB_ROWS = 344 rows (an odd multiple of 8) behind buffer 0, which flips bit 2 of (row >> 1) & 7
        const int lchunk = (lane & 7) ^ ((r >> 1) & 7) ^ ((c & 1) << 2);
        const bool ok = r < t_hpix && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        return ok ? (unsigned)((((t_n * a.H + iy) * a.W + ix) * a.Xs + lchunk * 8) * 2) : OOR;
    };
    auto setup_tile = [&]() {
    if constexpr (HALO) {
        const int pt = xcd * a.pt_per_xcd + ptl, full = a.tiles_x * a.tiles_y, per = full + a.tiles_strip;
        t_n = pt / per;
        const int rr = pt - t_n * per;
        if (rr < full) {
            // 16 x 16 tiles in BLOCKED order: bands of 4 tile rows, inside a band blocks of 8 tile columns, row-major inside a block.  The
            // 32 workgroups an XCD runs at a time then cover a compact 128 x 64-pixel region instead of one and a half tile rows, so most
            // halo rows / columns a tile shares with its neighbours are fetched while a neighbour holds them in the XCD's L2 (r03 PMC:
            // the row-major order re-fetched 1.29 x the algorithmic bytes from beyond L2)
            const int band_sz = a.tiles_x * 4, band = rr / band_sz, q = rr - band * band_sz;
            const int hb = a.tiles_y - band * 4 < 4 ? a.tiles_y - band * 4 : 4;   // tile rows of this band
            const int nfull = (a.tiles_x >> 3) * 8 * hb;                            // tiles of the band inside whole 8-column blocks
            int ty, tx;
            if (q < nfull) {
                const int blk = q / (8 * hb), o = q - blk * 8 * hb;
                ty = o >> 3; tx = blk * 8 + (o & 7);
            } else {
                const int remx = a.tiles_x & 7, o = q - nfull;
                ty = o / remx; tx = (a.tiles_x & ~7) + o - ty * remx;
            }
            t_y0 = (band * 4 + ty) * 16; t_x0 = tx * 16;
            t_twl = 4; t_hw = 18; t_magic = 3641; t_hpix = 324;
        } else {   // the bottom strip: 8 x 32 tiles
            t_y0 = a.tiles_y * 16; t_x0 = (rr - full) * 32;
            t_twl = 5; t_hw = 34; t_magic = 1928; t_hpix = 340;
        }
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int pl = MF ? wc * 64 + r4 * 16 + hp16 : wc * 64 + (r4 & 1) * 32 + hp32;   // tile-local pixel of this lane in the fragment
            hpb[r4] = (pl >> t_twl) * t_hw + (pl & ((1 << t_twl) - 1));
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int hB = q >> 1, i = q & 1;
        const int m = pix0 + (i * 2 + (wave >> 2)) * 64 + hB * 32 + (wave & 3) * 8 + (lane >> 3);
        p_base[q] = 0; p_taps[q] = 0u; p_base2[q] = OOR;
        if (a.x2 && m < a.M) {
            const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
            const int ho = r / a.Wo, wo = r - ho * a.Wo;
            p_base2[q] = (unsigned)((((n * a.H2 + ho * a.stride2) * a.W2 + wo * a.stride2) * a.Xs2) * 2 + chunk * 16);
        }
        if (a.pointwise) {
            if (m < a.M) { p_base[q] = m * a.Xs * 2 + chunk * 16; p_taps[q] = 1u; }
        } else if (m < a.M) {
            const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
            const int ho = r / a.Wo, wo = r - ho * a.Wo;
            const int hi0 = ho * a.stride - a.pad_top, wi0 = wo * a.stride - a.pad_left;
            p_base[q] = (((n * a.H + hi0) * a.W + wi0) * a.Xs) * 2 + chunk * 16;
            unsigned bits = 0u, bit = 1u;
            for (int dy = 0; dy < a.kh; ++dy)
                for (int dx = 0; dx < a.kw; ++dx, bit <<= 1)
                    if ((unsigned)(hi0 + dy) < (unsigned)a.H && (unsigned)(wi0 + dx) < (unsigned)a.W) bits |= bit;
            p_taps[q] = bits;
        }
    }
    };
    setup_tile();
    auto walk_next = [&](KWalk &w) {
        if (a.korder == 0) {
            w.cc0 += 8;
            if (w.cc0 == a.cpt) {
                w.cc0 = 0; ++w.tap;
                if (++w.kw == a.kw) { w.kw = 0; ++w.kh; }
            }
        } else {
            ++w.tap;
            if (++w.kw == a.kw) { w.kw = 0; ++w.kh; }
            if (w.tap == n_taps) { w.tap = 0; w.kh = 0; w.kw = 0; w.cc0 += 8; }
        }
    };
    auto stage_A = [&](int kt, int hA, int hbuf) {
        if (ABL == 1 && kt >= 2) return;
        char *dst = smem + ((kt & 1) * A_SLOTS + hbuf) * PP_HALF + wave * (8 * ROWB);
        const unsigned dead = (unsigned)((nk - 1 - kt) >> 31) << 31;  // 2^31 for the tiles past the end: out of range, zero fill
        const int ktc = kt < nk ? kt : nk - 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void *)(dst + i * 64 * ROWB), 16,
                                                     (int)((unsigned)(a_off + hA * a_half + i * a_pass) | dead), ktc * (BK * 2), 0, 0);
    };
    auto stage_B = [&](int kt, int hB, int hbuf, const KWalk &w) {
        if (ABL == 1 && kt >= 2) return;
        char *dst = smem + ((kt & 1) * 4 + hbuf) * PP_HALF + wave * (8 * ROWB);
        const unsigned dead = (unsigned)((nk - 1 - kt) >> 31) << 31;
        if (a.x2 && kt >= a.nk_a) {   // DUAL: K tiles past nk_a are 64-channel chunks of the second tensor's (strided) pixel
            const int ktc = kt < nk ? kt : nk - 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void *)(dst + i * 64 * ROWB), 16, (int)(p_base2[hB * 2 + i] | dead),
                                                         (ktc - a.nk_a) * (BK * 2), 0, 0);
            return;
        }
        const int soff = ((w.kh * a.W + w.kw) * a.Xs + w.cc0 * 8) * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned voff = ((p_taps[hB * 2 + i] >> (w.tap & 31)) & 1u) ? ((unsigned)(p_base[hB * 2 + i] + soff) | dead) : OOR;
            // (not non-temporal: a linear tile reads every activation line nine times -- once per tap -- and needs the L2 for eight of them;
            // r03 PMC with nt: 4.3 x the L2-miss traffic.  And no run-time switch here: a uniform branch around these DMAs split the
            // phase's basic block and cost the one-tile form 10 % -- found by a same-box comparison against the round-2 tree)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)(dst + i * 64 * ROWB), 16, (int)voff, 0, 0, 0);
        }
    };

    // HALO: piece (wave + 8 j) of channel chunk c's halo -> buffer c & 1; pieces past the 41st and chunks past the end are a zero fill of
    // the 1-KiB dummy area (same instruction count in every K tile).  j is wave-uniform: the select chain costs five v_cndmask.
    auto stage_H = [&](int c, int j) {
        if constexpr (HALO) {
            const int piece = wave + 8 * j;
            const bool real = j < 6 && piece < HB_ROWS / 8 && c < nch;
            const unsigned off = piece_off(j < 6 ? j : 5, c);
            char *dst = smem + (real ? HB0_OFF + (c & 1) * HB_BYTES + piece * 1024 : HB_DUMMY);
            // The halo is requested NON-TEMPORAL: every line is read once per tile (its neighbours' reads of the shared halo columns come within
            // microseconds), so it need not push the weights -- re-read by every workgroup of the XCD per K tile -- out of the XCD's L2.
            // r03 PMC (60 x 200 x 336, head form): L2-miss traffic 2.43 -> 2.27 GB per launch = 1.11 x the algorithmic bytes (1.28 x on linear
            // tiles, whose nine reads per line need the L2 and get 4.3 x the traffic under nt); step time unchanged (same-box A/B).
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void *)dst, 16, (int)(real ? off : OOR), (c < nch ? c : 0) * (BK * 2), 0, 2);
        }
    };

    f32x16 acc[MF ? 1 : 4][MF ? 1 : 2];
    f32x4 acc4[MF ? 8 : 1][MF ? 4 : 1];  // MF 1: [16-row fragment of the wave's 128 couts][16-pixel fragment of its 64 pixels]
    auto zero_acc = [&]() {
    if constexpr (MF == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc4[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    };
    zero_acc();

    int fa_off[4], fb_off[4];  // per k-step fragment offsets inside a half tile (MF 1 uses two: K 32 per step)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        fa_off[kk] = MF ? swz(wr * 64 + l16, (kk & 1) * 4 + lq) : swz(wr * 64 + lr, kk * 2 + lh);
        fb_off[kk] = MF ? swz(wc * 32 + l16, (kk & 1) * 4 + lq) : swz(wc * 32 + lr, kk * 2 + lh);
    }
    // HALO: the lane's k-chunk bits in address position
    const int hkq = MF ? (lq << 4) : (lh << 4);
    // operand registers of a phase: MF 0 fa[row fragment 0-1][k step 0-3], fb[pixel half][k step];
    //                               MF 1 fa[r][s] = row fragment (2r + (s >> 1)), k step (s & 1); fb[h][s] likewise per pixel half
    bf16x8 fa[2][4], fb[2][4];
    unsigned long long st[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, keep[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // ABL 4: s_memtime stamps of K tile nk/2
#define PP_STAMP(I) if (ABL == 4) { __builtin_amdgcn_sched_barrier(0); st[I] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }

    // The accumulators of a phase are threaded through empty volatile asm statements on both sides of its MFMA cluster:
    // MFMA builtins are pure, and without the pin the optimiser sinks them past the barriers into the next phase's
    // load section (seen in the ISA), which destroys the ping-pong.
#define PP_PIN(I0)                                                                                                  \
    if constexpr (MF == 0) {                                                                                        \
        asm volatile("" : "+v"(acc[I0][0]), "+v"(acc[I0][1]), "+v"(acc[I0 + 1][0]), "+v"(acc[I0 + 1][1]));          \
    } else {                                                                                                        \
        asm volatile("" : "+v"(acc4[2 * (I0)][0]), "+v"(acc4[2 * (I0)][1]), "+v"(acc4[2 * (I0)][2]), "+v"(acc4[2 * (I0)][3]),                 \
                          "+v"(acc4[2 * (I0) + 1][0]), "+v"(acc4[2 * (I0) + 1][1]), "+v"(acc4[2 * (I0) + 1][2]), "+v"(acc4[2 * (I0) + 1][3]), \
                          "+v"(acc4[2 * (I0) + 2][0]), "+v"(acc4[2 * (I0) + 2][1]), "+v"(acc4[2 * (I0) + 2][2]), "+v"(acc4[2 * (I0) + 2][3]), \
                          "+v"(acc4[2 * (I0) + 3][0]), "+v"(acc4[2 * (I0) + 3][1]), "+v"(acc4[2 * (I0) + 3][2]), "+v"(acc4[2 * (I0) + 3][3])); \
    }
#define PP_SYNC_LOADS(I0)                                                     \
    __builtin_amdgcn_sched_barrier(0);                                        \
    PP_STAMP((I0) * 5 / 2 + 1)                                                \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        \
    if (ABL != 1) { if (HALO) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); } \
    PP_STAMP((I0) * 5 / 2 + 2)                                                \
    __builtin_amdgcn_s_barrier();                                             \
    PP_STAMP((I0) * 5 / 2 + 3)                                                \
    PP_PIN(I0)                                                                \
    __builtin_amdgcn_sched_barrier(0);                                        \
    __builtin_amdgcn_s_setprio(1);
#define PP_END_PHASE(I0)                                                      \
    PP_PIN(I0)                                                                \
    __builtin_amdgcn_s_setprio(0);                                            \
    PP_STAMP((I0) * 5 / 2 + 4)                                                \
    __builtin_amdgcn_sched_barrier(0);                                        \
    __builtin_amdgcn_s_barrier();                                             \
    __builtin_amdgcn_sched_barrier(0);
#define PP_MFMA(I0)                                                                                                  \
    if constexpr (MF == 0) {                                                                                         \
        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                           \
            acc[I0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][kk], fb[0][kk], acc[I0][0], 0, 0, 0);         \
            acc[I0 + 1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][kk], fb[0][kk], acc[I0 + 1][0], 0, 0, 0); \
            acc[I0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0][kk], fb[1][kk], acc[I0][1], 0, 0, 0);         \
            acc[I0 + 1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1][kk], fb[1][kk], acc[I0 + 1][1], 0, 0, 0); \
        }                                                                                                            \
    } else {                                                                                                         \
        _Pragma("unroll") for (int k2 = 0; k2 < 2; ++k2)                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                \
            acc4[2 * (I0) + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i >> 1][(i & 1) * 2 + k2], fb[j >> 1][(j & 1) * 2 + k2], \
                                                                            acc4[2 * (I0) + i][j], 0, 0, 0);         \
    }

    // HEAD: the 16 x 256 head weights -> LDS (8 KiB above the bias copy), one 1-KiB DMA per wave; lane -> (row, physical
    // 16-B chunk), logical chunk = physical ^ row so that the 16 rows of an A-fragment read fall on 16 different slots
    constexpr int W2_OFF = HALO ? HB_W2 : PT * EP_STRIDE + CT * 4;
    if constexpr (HEAD) {
        __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void *)a.w2, 0, 16 * 256 * 2, 0x00020000);
        const int i = wave * 64 + lane, row = i >> 5, phys = i & 31;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_void *)(smem + W2_OFF + wave * 1024), 16, row * 512 + ((phys ^ row) & 31) * 16, 0, 0, 0);
    }
    // ---- prologue: tile 0 and the A0/B0/B1 halves of tile 1 (seven half tiles)
    KWalk w2 = {0, 0, 0, 0};
    auto issue_prologue = [&]() {
        if constexpr (HALO) {
            // chunk 0's halo, K tile 0's A halves, then "phase 1 of K tile -1" (A0 of K tile 1 + one dummy piece): from here on every phase
            // issues what it issues in the steady state, so one immediate (vmcnt 5 = the two youngest phases' pieces) serves every wait
#pragma unroll
            for (int j = 0; j < 6; ++j) stage_H(0, j);
            stage_A(0, 0, H_A0); stage_A(0, 1, H_A1);
            stage_A(1, 0, H_A0); stage_H(nch, 0);
            return;
        }
        KWalk w0 = {0, 0, 0, 0}, w1 = w0;
        walk_next(w1);
        w2 = w1;
        walk_next(w2);
        stage_A(0, 0, H_A0); stage_B(0, 0, H_B0, w0); stage_B(0, 1, H_B1, w0); stage_A(0, 1, H_A1);
        stage_A(1, 0, H_A0); stage_B(1, 0, H_B0, w1); stage_B(1, 1, H_B1, w1);
    };
    issue_prologue();
    bool first_tile = true;
    for (;;) {   // one pass unless PERS
    if (!PERS || first_tile) {
        // also retires the (older) bias load.  HALO: 13 pieces issued, the first 8 (halo + A0 / A1 of K tile 0) needed now
        if (HALO) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        if (tid < CT) bias_lds[tid] = bias_early;         // read in the epilogue, hundreds of barriers later
    } else {
        // behind the seven prologue half tiles (14 DMAs, the first 6 needed now) sit the 16 stores of the previous tile's epilogue
        if (HALO) asm volatile("s_waitcnt vmcnt(21)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // stagger: group 1 runs half a phase behind group 0
    __builtin_amdgcn_sched_barrier(0);

    unsigned long long clk0 = 0, rt0 = 0;
    if (ABL == 4) { clk0 = __builtin_readcyclecounter(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    int s_tap = 0, s_dy = 0, s_dx = 0, s_ct = 0;   // HALO: tap / channel chunk of K tile t (K = (chunk, tap, channel in chunk))
    for (int t = 0; t < nk; ++t) {
        const char *T = smem + (t & 1) * (A_SLOTS * PP_HALF);
        PP_STAMP(0)
        // phase 0: cout rows 0-63 of the wave x its 64 pixels
        if constexpr (HALO) {
            // B fragments out of the chunk's halo: row = the pixel's halo row shifted by the tap, 16-B chunk XOR (row >> 1) & 7 as staged.
            // The buffer's base (a multiple of 16 rows) rides in the row index: the swizzle bits are unchanged by it.
            const int s_row = s_dy * t_hw + s_dx + (HB0_OFF / ROWB) + (s_ct & 1) * HB_ROWS;
            if constexpr (MF == 1) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const int pr = hpb[r4] + s_row;
                    const int ad = (pr << 7) + ((((pr << 3) & 0x70)) ^ hkq);
                    fb[r4 >> 1][(r4 & 1) * 2 + 0] = *reinterpret_cast<const bf16x8 *>(smem + ad);
                    fb[r4 >> 1][(r4 & 1) * 2 + 1] = *reinterpret_cast<const bf16x8 *>(smem + (ad ^ 64));
                }
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pr = hpb[h] + s_row;
                    const int ad = (pr << 7) + ((((pr << 3) & 0x70)) ^ hkq);
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk) fb[h][kk] = *reinterpret_cast<const bf16x8 *>(smem + (ad ^ (kk << 5)));
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fb[0][kk] = *reinterpret_cast<const bf16x8 *>(T + H_B0 * PP_HALF + fb_off[kk] + (MF ? (kk >> 1) * 16 * ROWB : 0));
            fb[1][kk] = *reinterpret_cast<const bf16x8 *>(T + H_B1 * PP_HALF + fb_off[kk] + (MF ? (kk >> 1) * 16 * ROWB : 0));
        }
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = *reinterpret_cast<const bf16x8 *>(T + H_A0 * PP_HALF + fa_off[kk] + (MF ? (kk >> 1) * 16 * ROWB : 0));
            fa[1][kk] = *reinterpret_cast<const bf16x8 *>(T + H_A0 * PP_HALF + fa_off[kk] + 32 * ROWB + (MF ? (kk >> 1) * 16 * ROWB : 0));
        }
        stage_A(t + 1, 1, H_A1);
        PP_SYNC_LOADS(0)
        PP_MFMA(0)
        PP_END_PHASE(0)
        PP_STAMP(5)
        // phase 1: cout rows 64-127
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fa[0][kk] = *reinterpret_cast<const bf16x8 *>(T + H_A1 * PP_HALF + fa_off[kk] + (MF ? (kk >> 1) * 16 * ROWB : 0));
            fa[1][kk] = *reinterpret_cast<const bf16x8 *>(T + H_A1 * PP_HALF + fa_off[kk] + 32 * ROWB + (MF ? (kk >> 1) * 16 * ROWB : 0));
        }
        stage_A(t + 2, 0, H_A0);
        if constexpr (HALO) stage_H(s_ct + 1, s_tap);   // one piece of the NEXT chunk's halo per K tile (pieces 6-8: the dummy fill)
        else { stage_B(t + 2, 0, H_B0, w2); stage_B(t + 2, 1, H_B1, w2); }
        PP_SYNC_LOADS(2)
        PP_MFMA(2)
        PP_END_PHASE(2)
        PP_STAMP(10)
        if (ABL == 4 && t == nk / 2) {
#pragma unroll
            for (int i = 0; i < 11; ++i) keep[i] = st[i];
        }
        if constexpr (HALO) {
            if (++s_dx == 3) { s_dx = 0; ++s_dy; }
            if (++s_tap == 9) { s_tap = 0; s_dy = 0; ++s_ct; }
        } else walk_next(w2);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();  // every wave has now passed 2 + 4 nk barriers
    unsigned long long clk1 = 0, rt1 = 0;
    if (ABL == 4) { clk1 = __builtin_readcyclecounter(); rt1 = __builtin_amdgcn_s_memrealtime(); }
#undef PP_STAMP
#undef PP_PIN
#undef PP_SYNC_LOADS
#undef PP_END_PHASE
#undef PP_MFMA
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the zero-fill DMAs of the two tiles past the end
    __syncthreads();
    if constexpr (PERS) {
        // ---- persistent form: request the next tile's prologue NOW (every wave is past its last fragment read), then run this tile's
        // epilogue out of the registers through the wave's private slab -- builtin vector types only (see conv1x1_stream_kernel)
        // HALO: the tile's first pixel (its top-left corner); the wave's pixels are rows / columns of the 16 x 16 block behind it
        const int pix_cur = HALO ? (t_n * a.H + t_y0) * a.W + t_x0 : pix0;
        const int cur_y0 = t_y0, cur_x0 = t_x0, cur_twl = t_twl;
        ptl += pstep;
        const bool more = ptl < a.pt_per_xcd && xcd * a.pt_per_xcd + ptl < a.n_ptiles;
        if (more) {
            pix0 = (xcd * a.pt_per_xcd + ptl) * PT;
            setup_tile();
            issue_prologue();
        }
        // (HALO: the slabs sit in halo buffer 1 -- its last reader was a K tile of this tile, its next writer is a piece issued inside the
        // next tile's loop, behind that tile's opening barrier)
        char *slab = smem + (HALO ? HB0_OFF + HB_BYTES : 8 * 128 * ROWB + CT * 4) + wave * 2560;
        const long long rem = ((long long)a.M - pix_cur) * a.Ctot * 2 - (a.c_off + cout0) * 2;
        __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)(a.y + (size_t)pix_cur * a.Ctot + a.c_off + cout0), 0,
                                                                        (int)(rem > 0x7fffffffLL ? 0x7fffffffLL : rem), 0x00020000);
        // a slab holds 16 pixels x 64 output channels (144-B rows): its read-out is 8 lanes per pixel = one whole 128-B line per pixel and
        // store instruction.  (32 x 32 slabs stored 64-B half lines: rocprofv3 FETCH_SIZE showed 15 % more HBM reads -- the L2 fills a
        // partially written line first.)
        const int e_px = lane >> 3, e_ch = lane & 7;
        const int v_io = ((wc * 64 + e_px) * a.Ctot + wr * 128 + e_ch * 8) * 2, row_b = a.Ctot * 2;
        auto flush_slab = [&](int px0, int c0) {   // pixels px0 .. px0 + 15 of the wave's 64, couts c0 .. c0 + 63 of its 128
            // compiler-level ordering points on both sides of the read-out: the slab is written by some lanes and read back by others in
            // another vector type, and hipcc may otherwise move (or duplicate into the non-writing lanes) the reads across the writes --
            // seen in bottleneck64_kernel's slab epilogue (r03).  LDS operations of one wave execute in issue order: no hardware wait.
            MD_WAVE_LDS_ORDER();
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                u32x4 v = *reinterpret_cast<const u32x4 *>(slab + (e_px + 8 * it) * 144 + e_ch * 16);
                if constexpr (HALO) {   // tile-local pixel -> (row, column) by the tile's shape; outside the image: dropped by the range check
                    const int pl = wc * 64 + px0 + 8 * it + e_px;
                    const int row = pl >> cur_twl, col = pl & ((1 << cur_twl) - 1);
                    const bool ok = cur_y0 + row < a.H && cur_x0 + col < a.W;
                    const unsigned off = ok ? (unsigned)(((row * a.W + col) * a.Ctot + wr * 128 + e_ch * 8) * 2) : OOR;
                    MD_BUFFER_STORE_B128(v, rs_y, off, c0 * 2, 2);
                } else
                MD_BUFFER_STORE_B128(v, rs_y, v_io + (px0 + 8 * it) * row_b, c0 * 2, 2);
            }
            MD_WAVE_LDS_ORDER();
        };
        if constexpr (MF == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {   // 32 x 32 accumulator tiles (ip * 2, ip * 2 + 1) x j, pixel half h (lanes lr >> 4 == h)
#pragma unroll
                        for (int ii = 0; ii < 2; ++ii) {
                            const int i = 2 * ip + ii;
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + wr * 128 + i * 32 + 8 * g + 4 * lh);
                                f32x2 s01 = (f32x2){acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]} + (f32x2){bv.x, bv.y};
                                f32x2 s23 = (f32x2){acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} + (f32x2){bv.z, bv.w};
                                if (GEN == 2) { s01.x = silu(s01.x); s01.y = silu(s01.y); s23.x = silu(s23.x); s23.y = silu(s23.y); }
                                u32x2 pk;
                                pk.x = pk_bf16(s01.x, s01.y);
                                pk.y = pk_bf16(s23.x, s23.y);
                                if (GEN == 0 && a.relu == 1) { pk.x = pk_relu_bf16(pk.x); pk.y = pk_relu_bf16(pk.y); }
                                if ((lr >> 4) == h) *reinterpret_cast<u32x2 *>(slab + (lr & 15) * 144 + ii * 64 + 16 * g + 8 * lh) = pk;
                            }
                        }
                        flush_slab(j * 32 + h * 16, ip * 64);
                    }
        } else {
#pragma unroll
            for (int j4 = 0; j4 < 4; ++j4)
#pragma unroll
                for (int q = 0; q < 2; ++q) {       // 16 x 16 accumulator tiles (4 q .. 4 q + 3) x j4
#pragma unroll
                    for (int ii = 0; ii < 4; ++ii) {
                        const f32x4 bv = *reinterpret_cast<const f32x4 *>(bias_lds + wr * 128 + (4 * q + ii) * 16 + 4 * lq);
                        const f32x4 v = acc4[4 * q + ii][j4];
                        float v0 = v[0] + bv.x, v1 = v[1] + bv.y, v2 = v[2] + bv.z, v3 = v[3] + bv.w;
                        if (GEN == 0 && a.relu == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        else if (GEN == 2) { v0 = silu(v0); v1 = silu(v1); v2 = silu(v2); v3 = silu(v3); }
                        u32x2 pk;
                        pk.x = pk_bf16(v0, v1);
                        pk.y = pk_bf16(v2, v3);
                        *reinterpret_cast<u32x2 *>(slab + hp16 * 144 + (ii * 16 + 4 * lq) * 2) = pk;
                    }
                    flush_slab(j4 * 16, q * 64);
                }
        }
        if (!more) return;
        zero_acc();
        first_tile = false;
        continue;
    }

    // ---- epilogue: bias (+act) -> bf16x4 -> LDS [pixel][cout] -> (+residual, ReLU) -> 16-B NHWC stores
    constexpr int CPP = CT / 8, EP_ITERS = PT * CPP / NT;
    char *E = smem;
    if constexpr (MF == 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c_local = wr * 128 + i * 16 + 4 * lq;  // 4 consecutive couts
            const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + c_local);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p_local = wc * 64 + j * 16 + hp16;
                float v0 = acc4[i][j][0] + bv.x, v1 = acc4[i][j][1] + bv.y, v2 = acc4[i][j][2] + bv.z, v3 = acc4[i][j][3] + bv.w;
                if (a.relu == 1 && !a.res) {
                    v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
                } else if (GEN == 2 || (GEN == 1 && a.relu == 2)) {
                    v0 = silu(v0); v1 = silu(v1); v2 = silu(v2); v3 = silu(v3);
                }
                uint2 pk;
                pk.x = pk_bf16(v0, v1);
                pk.y = pk_bf16(v2, v3);
                *reinterpret_cast<uint2 *>(E + p_local * EP_STRIDE + c_local * 2) = pk;
            }
        }
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int c_local = wr * 128 + i * 32 + 8 * g + 4 * lh;
            const float4 bv = *reinterpret_cast<const float4 *>(bias_lds + c_local);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p_local = wc * 64 + j * 32 + hp32;
                // packed adds (v_pk_add_f32), one-instruction bf16 pack, ReLU on the packed pair: 8 VALU per 4 values
                f32x2 s01 = (f32x2){acc[i][j][4 * g + 0], acc[i][j][4 * g + 1]} + (f32x2){bv.x, bv.y};
                f32x2 s23 = (f32x2){acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]} + (f32x2){bv.z, bv.w};
                if (GEN == 2 || (GEN == 1 && a.relu == 2)) { s01.x = silu(s01.x); s01.y = silu(s01.y); s23.x = silu(s23.x); s23.y = silu(s23.y); }
                uint2 pk;
                pk.x = pk_bf16(s01.x, s01.y);
                pk.y = pk_bf16(s23.x, s23.y);
                if (a.relu == 1 && !a.res) { pk.x = pk_relu_bf16(pk.x); pk.y = pk_relu_bf16(pk.y); }
                *reinterpret_cast<uint2 *>(E + p_local * EP_STRIDE + c_local * 2) = pk;
            }
        }
    }
    __syncthreads();
    if constexpr (HEAD) {
        // ---- fused head: y2[p][c2] = b2[c2] + sum_k W2[c2][k] * relu(conv)[p][k], K = 256 straight from the epilogue image.
        // v_mfma_f32_16x16x32_bf16 with A = W2 (16 head channels), B = 16 pixels; each wave owns 32 pixels.
        const int l16 = lane & 15, lq = lane >> 4;
        f32x4 h[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const bf16x8 wa = *reinterpret_cast<const bf16x8 *>(smem + W2_OFF + l16 * 512 + (((k2 * 4 + lq) ^ l16) & 31) * 16);
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 xb = *reinterpret_cast<const bf16x8 *>(E + (wave * 32 + f * 16 + l16) * EP_STRIDE + (k2 * 32 + lq * 8) * 2);
                h[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb, h[f], 0, 0, 0);
            }
        }
        const float4 hb = *reinterpret_cast<const float4 *>(a.b2 + 4 * lq);
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            int m = pix0 + wave * 32 + f * 16 + l16;
            if constexpr (HALO) {   // tile-local pixel wave * 32 + f * 16 + l16
                const int pl = wave * 32 + f * 16 + l16;
                const int yy = t_y0 + (pl >> t_twl), xx = t_x0 + (pl & ((1 << t_twl) - 1));
                m = (yy < a.H && xx < a.W) ? (t_n * a.H + yy) * a.W + xx : a.M;
            }
            if (m >= a.M) continue;
            uint2 pk;
            pk.x = pk_bf16(h[f][0] + hb.x, h[f][1] + hb.y);
            pk.y = pk_bf16(h[f][2] + hb.z, h[f][3] + hb.w);
            *reinterpret_cast<uint2 *>(a.y2 + (size_t)m * 16 + 4 * lq) = pk;
        }
        return;
    }
    auto out_offset = [&](int m, int c) -> size_t {
        if (!(GEN == 1 && a.adv)) return (size_t)m * a.Ctot + a.c_off + c;   // plain, or a channel-concat output
        const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
        const int ho = r / a.Wo, wo = r - ho * a.Wo;
        return (((size_t)n * a.Hf + ho * a.os + a.oy) * a.Wf + wo * a.os + a.ox) * a.Ctot + a.c_off + c;
    };
    auto res_offset = [&](int m, int c) -> size_t {
        if (a.Rs) return (size_t)m * a.Rs + c;
        if (!(GEN == 1 && a.res_up)) return out_offset(m, c);
        const int n = m / (a.Ho * a.Wo), r = m - n * (a.Ho * a.Wo);
        const int ho = r / a.Wo, wo = r - ho * a.Wo;
        const int Hr = (a.Ho + 1) >> 1, Wr = (a.Wo + 1) >> 1;
        return (((size_t)n * Hr + (ho >> 1)) * Wr + (wo >> 1)) * a.Cout + c;
    };
    auto tile_pixel = [&](int p_local) -> int {   // output pixel index of the tile's pixel p_local (a.M: outside the image)
        if constexpr (HALO) {
            const int yy = t_y0 + (p_local >> t_twl), xx = t_x0 + (p_local & ((1 << t_twl) - 1));
            return (yy < a.H && xx < a.W) ? (t_n * a.H + yy) * a.W + xx : a.M;
        }
        return pix0 + p_local;
    };
    u32x4 rres[EP_ITERS];
    if (a.res) {
#pragma unroll
        for (int it = 0; it < EP_ITERS; ++it) {
            const int e = tid + it * NT;
            const int m = tile_pixel(e / CPP), c = cout0 + (e % CPP) * 8;
            rres[it] = (u32x4){0u, 0u, 0u, 0u};
            if (m < a.M) rres[it] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(a.res + res_offset(m, c)));
        }
    }
#pragma unroll
    for (int it = 0; it < EP_ITERS; ++it) {
        const int e = tid + it * NT;
        const int p_local = e / CPP, cc = e % CPP;
        const int m = tile_pixel(p_local), c = cout0 + cc * 8;
        if (m >= a.M) continue;
        u32x4 v = *reinterpret_cast<const u32x4 *>(E + p_local * EP_STRIDE + cc * 16);
        if (a.res) {
            const u32x4 rv = rres[it];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x2 sum = (f32x2){__uint_as_float(v[k] << 16), __uint_as_float(v[k] & 0xffff0000u)} +
                                  (f32x2){__uint_as_float(rv[k] << 16), __uint_as_float(rv[k] & 0xffff0000u)};
                v[k] = pk_bf16(sum.x, sum.y);
                if (a.relu == 1) v[k] = pk_relu_bf16(v[k]);
            }
        }
        if (ABL == 2 && v[0] != 0x12345u) continue;
        __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(a.y + out_offset(m, c)));
    }
    if (ABL == 4 && a.dbg && blockIdx.x == (gridDim.x / 2 & ~7u)) {  // MD_DIAG build only: a mid-grid workgroup's stamps -> the stamp buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long clk_end = __builtin_readcyclecounter();
        __syncthreads();
        if (lane == 0) {
            unsigned long long *dbg = a.dbg + wave * 16;
            for (int i = 0; i < 11; ++i) dbg[i] = keep[i];
            dbg[11] = clk1 - clk0; dbg[12] = rt1 - rt0;
            dbg[13] = clk0 - clk_start; dbg[14] = clk_end - clk1;  // prologue / epilogue cycles
        }
    }
    return;
    }   // tile loop
}

// The MFMA shape of the ping-pong kernel per layer (r02, tools/pp_mf_ab.py, batch 60, same box, bit-identical results): the chip
// holds a higher clock on v_mfma_f32_16x16x32_bf16 in the long K = 2304 loops of the big 3x3 layers (200x336: +2.5 %, 100x168:
// +1.6 %, 50x84: +0.9 %) and a lower one on the short-K / small-grid layers (1x1 1024->256: -7 %, 3x3 512->512 at 25x42: -3 %).
static bool pingpong_wants_16x16(const ConvArgs &a) { return a.kh == 3 && a.kw == 3 && a.Kpad >= 2304 && a.M >= 400000; }

// The HALO form's tile map: 16 x 16-pixel tiles, row-major per image (neighbouring tiles -- which share halo columns -- sit on one XCD).
static long long pingpong_halo_tiles_per_image(int H, int W, int *tiles_y, int *tiles_strip) {
    const int rem = H % 16;
    const bool strip = rem > 0 && rem <= 8;   // at most 8 rows left under the 16-row tiles: a strip of 8 x 32 tiles instead of half-empty 16 x 16 ones
    const int ty = strip ? H / 16 : (H + 15) / 16, ts = strip ? (W + 31) / 32 : 0;
    if (tiles_y) *tiles_y = ty;
    if (tiles_strip) *tiles_strip = ts;
    return (long long)ty * ((W + 15) / 16) + ts;
}
static void pingpong_halo_tiles(ConvArgs &a) {
    a.tiles_x = (a.W + 15) / 16;
    a.n_ptiles = (int)(a.N * pingpong_halo_tiles_per_image(a.H, a.W, &a.tiles_y, &a.tiles_strip));
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
}
// Preconditions of the HALO form (checked by the dispatcher besides the ping-pong kernel's own): 3x3 / stride 1 / pad 1 with korder-1 weights.
static bool pingpong_halo_takes(const ConvArgs &a) {
    return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad_top == 1 && a.pad_left == 1 && a.korder == 1 && a.Cin % 64 == 0 && a.Ho == a.H &&
           a.Wo == a.W && !a.x2 && !a.res_up && a.x_bytes != 0 && (long long)a.N * pingpong_halo_tiles_per_image(a.H, a.W, nullptr, nullptr) < 0x7fffffffLL / 8 &&
           (long long)(15 * a.W + 32) * a.Ctot * 2 < 0x7fffffffLL;   // the persistent form's 32-bit store offsets inside a tile
}

template <int MF, bool HALO>
static int launch_conv_pingpong_head_mf(ConvArgs &a, hipStream_t s, long long blocks, int lds) {
    auto k = conv_pingpong_kernel<0, MF, 0, true, false, HALO>;
    if (ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

static int launch_conv_pingpong_head(ConvArgs &a, hipStream_t s, bool halo) {
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_PINGPONG;
    a.n_ctiles = 1;
    a.n_ptiles = (a.M + 255) / 256;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    if (halo) pingpong_halo_tiles(a);
    const long long blocks = (long long)a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    if (halo) return pingpong_wants_16x16(a) ? launch_conv_pingpong_head_mf<1, true>(a, s, blocks, HB_LDS_HEAD) : launch_conv_pingpong_head_mf<0, true>(a, s, blocks, HB_LDS_HEAD);
    const int lds = 256 * (256 * 2 + 16) + 256 * 4 + 16 * 256 * 2;  // epilogue image + bias + head weights
    return pingpong_wants_16x16(a) ? launch_conv_pingpong_head_mf<1, false>(a, s, blocks, lds) : launch_conv_pingpong_head_mf<0, false>(a, s, blocks, lds);
}

// the persistent form: one workgroup per CU, S = 32 / n_ctiles workgroups per (XCD, cout tile) stride through the XCD's pixel range
template <int MF, bool HALO = false>
static int launch_conv_pingpong_pers(ConvArgs &a, hipStream_t s) {
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_PINGPONG;
    a.n_ctiles = a.Cout / 256;
    a.n_ptiles = (a.M + 255) / 256;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    if (HALO) pingpong_halo_tiles(a);
    const int lds = HALO ? HB_LDS : 8 * 128 * ROWB + 256 * 4 + 8 * 2560;   // staging buffers + bias + eight wave-private slabs (HALO: in halo buffer 1)
    auto k = a.relu == 2 ? conv_pingpong_kernel<0, MF, 2, false, true, HALO> : conv_pingpong_kernel<0, MF, 0, false, true, HALO>;
    if (ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

static int launch_conv_dual_pingpong(ConvArgs &a, hipStream_t s) {
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_PINGPONG;
    a.n_ctiles = a.Cout / 256;
    a.n_ptiles = (a.M + 255) / 256;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    const long long blocks = (long long)a.n_ctiles * a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    const int lds = 256 * (256 * 2 + 16) + 256 * 4;
    auto k = conv_pingpong_kernel<0, 0, 0>;
    if (ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

// plain (one tile per workgroup) HALO form: GEN 0 / 2 epilogues, optional residual with the output's layout or a channel slice
template <int MF>
static int launch_conv_pingpong_halo(ConvArgs &a, hipStream_t s) {
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_PINGPONG;
    a.n_ctiles = a.Cout / 256;
    pingpong_halo_tiles(a);
    const long long blocks = (long long)a.n_ctiles * a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    auto k = a.relu == 2 ? conv_pingpong_kernel<0, MF, 2, false, false, true> : conv_pingpong_kernel<0, MF, 0, false, false, true>;
    if (ensure_dyn_lds((const void *)k, HB_LDS) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), HB_LDS, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

// (r04: conv_w128_kernel -- this tile on four waves with a 128 x 128 register tile each -- was built, bit-identical, and measured: it ties
// this kernel on dense data and loses 4-35 % on zero operands; profiles/r04_w128_experiment.txt, git history has the source.)
template <int ABL = 0, int MF = 0>
static int launch_conv_pingpong(ConvArgs &a, hipStream_t s) {
    ++g_launch_count;
    g_last_kernel = MD_CONV_KERNEL_PINGPONG;
    a.n_ctiles = a.Cout / 256;
    a.n_ptiles = (a.M + 255) / 256;
    a.pt_per_xcd = (a.n_ptiles + 7) / 8;
    const long long blocks = (long long)a.n_ctiles * a.pt_per_xcd * 8;
    if (blocks > 0x7fffffffLL) return MD_ERR_SIZE;
    const int lds = 256 * (256 * 2 + 16) + 256 * 4;  // 136,192 B: the epilogue image (>= the 128 KiB of staging buffers) + bias
    constexpr bool HAS_PLAIN = ABL == 0 || (ABL == 4 && MF == 0);
    const bool cat_only = a.adv && a.os == 1 && !a.oy && !a.ox && a.Ho == a.Hf && a.Wo == a.Wf;
    const bool plain = HAS_PLAIN && (!a.adv || cat_only) && !a.res_up;
    auto k = conv_pingpong_kernel<ABL, MF, 1>;
    if (plain) k = a.relu == 2 ? conv_pingpong_kernel<ABL, MF, HAS_PLAIN ? 2 : 1> : conv_pingpong_kernel<ABL, MF, HAS_PLAIN ? 0 : 1>;
    if (ensure_dyn_lds((const void *)k, lds) != MD_OK) return MD_ERR_HIP;
    hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(512), lds, s, a);
    return hipGetLastError() == hipSuccess ? MD_OK : MD_ERR_HIP;
}

}  // namespace md

using namespace md;

// Required weight padding for a given Cout (the tile the dispatcher will pick): exported so the
// host packer pads consistently.
extern "C" int md_conv2d_last_kernel(void) { return g_last_kernel; }
extern "C" long long md_conv2d_launch_count(void) { return g_launch_count; }
namespace md {
void md_note_conv_kernel(int id) { g_last_kernel = id; ++g_launch_count; }
}

extern "C" int md_conv2d_cout_tile(int cout) { return cout > 64 ? 128 : (cout > 32 ? 64 : 32); }

struct HeadArgs {   // the fused RPN head: y2 has 16 channels, y is not written
    const uint16_t *w2;
    const float *b2;
    uint16_t *y2;
};
#ifdef MD_DIAG
static unsigned long long *g_stamp_buf = nullptr;
// device buffer (>= 8 * 16 * 8 bytes) that receives the cycle stamps of the variant 19 / 25 launches
extern "C" int md_diag_set_stamp_buffer(void *p) { g_stamp_buf = (unsigned long long *)p; return MD_OK; }
#endif
#define MD_ERR_UNSUPPORTED_INTERNAL 100  // conv2d_entry with a head on a layer the fused kernel does not take

static int conv2d_entry(MD_AOT_ARGS, const HeadArgs *head) {
    if (nparam != 5) return MD_ERR_NPARAM;
    if (!params || !extra || !params[1] || !params[2]) return MD_ERR_ARG;  // x / y may be null for an empty batch
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16") || !dtype_is(dtypes, 2, "float32") ||
        !dtype_is(dtypes, 4, "bfloat16"))
        return MD_ERR_ARG;
    if (!ndims || !shapes || ndims[0] != 4 || ndims[1] != 2 || ndims[4] != 4) return MD_ERR_ARG;
    const md_conv2d_attrs *at = (const md_conv2d_attrs *)extra;
    if (at->reserved0 != 0) return MD_ERR_ARG;
    const Tune tn = resolve_tune(&at->tune);
    const long long chunk_limit = tn.chunk_limit;   // per call (md_conv_tune.chunk_limit)
    // The LDS-DMA kernels address the activation tensor with 32-bit byte offsets.  A batch whose input (or output /
    // residual, which the kernels address with 64-bit math but the same image split applies to) exceeds 2 GiB is run
    // as consecutive image chunks on the same stream: every tensor of the call is sliced along N.
    {
        const long long n_img = shapes[0][0];
        const long long x_img = shapes[0][1] * shapes[0][2] * shapes[0][3] * 2;
        if (n_img > 1 && x_img > 0 && x_img < chunk_limit && n_img * x_img >= chunk_limit && shapes[4][0] == n_img &&
            (!params[3] || (ndims[3] == 4 && shapes[3][0] == n_img))) {
            const long long per_max = chunk_limit / x_img;                 // images a chunk may hold (>= 1)
            const long long n_chunks = (n_img + per_max - 1) / per_max;
            const long long per = (n_img + n_chunks - 1) / n_chunks;          // even split: no tiny last chunk
            const long long y_img = shapes[4][1] * shapes[4][2] * shapes[4][3] * 2;
            const long long r_img = params[3] ? shapes[3][1] * shapes[3][2] * shapes[3][3] * 2 : 0;
            for (long long n0 = 0; n0 < n_img; n0 += per) {
                const long long nn = n_img - n0 < per ? n_img - n0 : per;
                int64_t sx[4] = {nn, shapes[0][1], shapes[0][2], shapes[0][3]};
                int64_t sy[4] = {nn, shapes[4][1], shapes[4][2], shapes[4][3]};
                int64_t sr[4] = {nn, 0, 0, 0};
                if (params[3]) { sr[1] = shapes[3][1]; sr[2] = shapes[3][2]; sr[3] = shapes[3][3]; }
                int64_t *sh2[5] = {sx, shapes[1], shapes[2], params[3] ? sr : shapes[3], sy};
                void *p2[5] = {(char *)params[0] + n0 * x_img, params[1], params[2],
                               params[3] ? (void *)((char *)params[3] + n0 * r_img) : nullptr, (char *)params[4] + n0 * y_img};
                HeadArgs h2;
                if (head) { h2 = *head; h2.y2 += n0 * shapes[4][1] * shapes[4][2] * 16; }
                const int rc = conv2d_entry(nparam, p2, ndims, sh2, dtypes, stream, extra, head ? &h2 : nullptr);
                if (rc != MD_OK) return rc;
            }
            return MD_OK;
        }
    }
    int variant_override = -1;
    ConvArgs a = {};   // (x2 == null: not a dual launch)
    a.x = (const uint16_t *)params[0];
    a.w = (const uint16_t *)params[1];
    a.bias = (const float *)params[2];
    a.res = (const uint16_t *)params[3];
    a.y = (uint16_t *)params[4];
    a.N = (int)shapes[0][0]; a.H = (int)shapes[0][1]; a.W = (int)shapes[0][2]; a.Cin = (int)shapes[0][3];
    // channel-slice operands: x = channels [x_c_off, x_c_off + x_cin) of the [N,H,W,Xs] tensor; residual = channels
    // [res_c_off, res_c_off + Cout) of a [N,Ho,Wo,Rs] tensor (res_slice)
    a.Xs = a.Cin; a.Rs = 0;
    if (at->x_cin > 0) {
        if (at->x_c_off < 0 || at->x_c_off % 8 || at->x_cin % 8 || at->x_c_off + at->x_cin > a.Xs) return MD_ERR_ARG;
        a.Cin = at->x_cin;
        if (a.x) a.x += at->x_c_off;
    } else if (at->x_c_off != 0) return MD_ERR_ARG;
    a.Hf = (int)shapes[4][1]; a.Wf = (int)shapes[4][2]; a.Ctot = (int)shapes[4][3];
    a.kh = at->kh; a.kw = at->kw; a.stride = at->stride; a.pad = at->pad; a.relu = at->relu;
    if (a.kh < 1 || a.kw < 1 || a.stride < 1 || a.pad < 0 || a.relu < 0 || a.relu > 2) return MD_ERR_ARG;
    a.adv = at->adv != 0;
    a.korder = at->korder;
    a.res_up = at->res_upsample != 0 && params[3] != nullptr;
    // one LDS staging buffer by default: measured r01 (tools/conv_ab.py), 4 resident workgroups per CU with a serial
    // DMA -> MFMA loop beat 2 double-buffered ones on every benchmark layer (+8...43 %); variant 2 keeps the double buffer
    // (every "auto, except ..." code counts as auto here: r03's first A/Bs of 34 / 35 / 39 / 40 ran their arm on the double-buffered loop
    // and read 1.5 ms per step too slow)
    a.single_buf = at->variant == 0 || at->variant == 20 || at->variant == 25 || (at->variant >= 30 && at->variant <= 40);
    a.stamp = 0; a.dbg = nullptr;
    if ((at->variant >= 17 && at->variant <= 19) || at->variant == 25 || at->variant == 26) {
        // timing ablations / stamp builds: wrong results by construction, so not part of the product library
        if (!kDiag) return MD_ERR_ARG;
#ifdef MD_DIAG
        a.stamp = at->variant == 25; a.dbg = g_stamp_buf;
#endif
    }
    if (at->variant == 25) variant_override = 2;
    if (at->variant == 20) variant_override = 2;
    if (a.res_up && a.adv) return MD_ERR_ARG;
    if (!a.adv) {
        a.Ho = a.Hf; a.Wo = a.Wf; a.Cout = a.Ctot;
        a.pad_top = a.pad_left = a.pad; a.os = 1; a.oy = a.ox = a.c_off = 0;
        if (a.Ho != (a.H + 2 * a.pad - a.kh) / a.stride + 1 || a.Wo != (a.W + 2 * a.pad - a.kw) / a.stride + 1)
            return MD_ERR_ARG;
    } else {
        a.Ho = at->sub_h; a.Wo = at->sub_w; a.Cout = at->cout;
        a.pad_top = at->pad_top; a.pad_left = at->pad_left;
        a.os = at->out_stride; a.oy = at->out_off_y; a.ox = at->out_off_x; a.c_off = at->c_off;
        if (a.Ho < 1 || a.Wo < 1 || a.os < 1 || a.oy < 0 || a.ox < 0 || a.c_off < 0 || a.c_off % 8 || a.Cout < 8) return MD_ERR_ARG;
        if (a.pad_top < 0 || a.pad_left < 0 || a.pad_top > 16384 || a.pad_left > 16384) return MD_ERR_ARG;
        // every written element must lie inside the output tensor
        if ((a.Ho - 1) * a.os + a.oy >= a.Hf || (a.Wo - 1) * a.os + a.ox >= a.Wf || a.c_off + a.Cout > a.Ctot) return MD_ERR_ARG;
    }
    if (a.Cin % 8 || a.Cout % 8 || a.Ctot % 8 || shapes[4][0] != a.N) return MD_ERR_ARG;
    a.Kreal = a.kh * a.kw * a.Cin;
    a.Kpad = (int)shapes[1][1];
    const int ctile = md_conv2d_cout_tile(a.Cout);
    const int cout_pad = (a.Cout + ctile - 1) / ctile * ctile;
    if (a.Kpad % BK || a.Kpad < a.Kreal || shapes[1][0] != cout_pad) return MD_ERR_ARG;
    if (numel(ndims, shapes, 2) != cout_pad) return MD_ERR_ARG;
    if (at->res_slice) {
        if (!params[3] || a.res_up || ndims[3] != 4 || shapes[3][0] != a.N || shapes[3][1] != a.Ho || shapes[3][2] != a.Wo ||
            a.os != 1 || a.oy || a.ox || a.Ho != a.Hf || a.Wo != a.Wf || at->res_c_off < 0 || at->res_c_off % 8 ||
            at->res_c_off + a.Cout > shapes[3][3])
            return MD_ERR_ARG;
        a.Rs = (int)shapes[3][3];
        a.res += at->res_c_off;
    } else if (params[3] && !a.res_up && (ndims[3] != 4 || numel(ndims, shapes, 3) != numel(ndims, shapes, 4))) return MD_ERR_ARG;
    if (a.res_up && (ndims[3] != 4 || shapes[3][0] != a.N || shapes[3][1] != (a.Ho + 1) / 2 || shapes[3][2] != (a.Wo + 1) / 2 ||
                     shapes[3][3] != a.Cout))
        return MD_ERR_ARG;
    const long long M = (long long)a.N * a.Ho * a.Wo;
    if (M <= 0) return MD_OK;
    if (!params[0] || !params[4]) return MD_ERR_ARG;
    if (M > 0x7fffffffLL || (long long)a.N * a.H * a.W > 0x7fffffffLL / 2 || a.H > 32000 || a.W > 32000)
        return MD_ERR_SIZE;
    a.M = (int)M;
    a.cpt = a.Cin / 8;
    a.pointwise = a.kh == 1 && a.kw == 1 && a.stride == 1 && a.pad_top == 0 && a.pad_left == 0 && a.H == a.Ho && a.W == a.Wo;
    hipStream_t s = (hipStream_t)stream;
    // variant: 0 = auto (cost model below); 1 = register-staged 128x128; 2 = LDS-DMA 128x128 with two staging buffers;
    // 11 = 128-cout halo kernel; 15 / 22 = ping-pong kernel (32x32x16 / 16x16x32 MFMA);
    // 20 = LDS-DMA 128x128 with one staging buffer; 27 = 64-cout halo kernel; 30 = conv1x1_stream_kernel where it applies,
    // 31 = auto without it; 17-19 / 25 = timing ablations / stamps, MD_DIAG
    // builds only (the product library answers MD_ERR_ARG)
    int variant = variant_override >= 0 ? variant_override : at->variant;
    const long long x_bytes = (long long)a.N * a.H * a.W * a.Xs * 2 - (at->x_cin > 0 ? at->x_c_off * 2 : 0), w_bytes = (long long)cout_pad * a.Kpad * 2;
    const bool dma_ok = x_bytes < 0x7fff0000LL && w_bytes < 0x7fff0000LL;  // 32-bit DMA offsets, out-of-range marker 2^31
    a.x_bytes = (unsigned)(dma_ok ? x_bytes : 0);
    a.w_bytes = (unsigned)(dma_ok ? w_bytes : 0);
    if (!dma_ok) variant = 1;
    if (a.korder != 0 && (a.korder != 1 || variant == 1 || a.Cin % 64 || a.kh * a.kw > 32)) return MD_ERR_ARG;
    // weight-stationary streaming kernel for pointwise layers with K <= 512 (variant 30 forces it where it applies)
    // auto: where it measured faster than the 128x128 kernel INSIDE the Faster R-CNN step (r02 tools/stream_insitu_tune.py, batch 60;
    // a replay loop on one layer flatters it: the activation tensor then survives in the Infinity Cache between launches):
    // 256->1024 + residual -0.40 ms/step (5 launches), 512->256 -0.12, 512->2048 + residual -0.09; 128->512 + residual +0.15 (stays on
    // the 128x128 kernel); the 128-cout forms lose 2-15 % -- and every workgroup gets at least 8 tiles to stream past its weights
    const bool no_stream = variant == 31;   // 31 = the dispatcher's choice without conv1x1_stream_kernel (A/B)
    const bool no_pers = variant == 33;     // 33 = the dispatcher's choice without the persistent form of the ping-pong kernel (A/B)
    const bool force_halo = variant == 34;  // 34 = the dispatcher's choice, with the HALO form of the ping-pong kernel wherever it applies
    const bool no_halo = variant == 35;     // 35 = the dispatcher's choice without the HALO form (A/B)
    const bool halo_fit = variant == 39;    // 39 = auto, with the HALO form also for the persistent / one-tile forms where the tiles fit the image (A/B)
    const bool halo_not_pers = variant == 40;   // 40 = auto, but layers the persistent form would take run on the one-tile HALO form where the tiles fit (A/B)
    if (no_stream || no_pers || force_halo || no_halo || halo_fit || halo_not_pers) variant = 0;
    const bool stream_auto = variant == 0 && !no_stream && !head && dma_ok && stream1x1_takes(a) && a.Cout % 256 == 0 && a.Cin != 128 &&
                             (M + 31) / 32 * (a.Cout / (a.Cin == 512 ? 128 : 256)) >= 4096;
    if ((variant == 30 || stream_auto) && !head) {
        const int rc = dma_ok ? launch_conv1x1_stream(a, s, tn) : MD_ERR_UNSUPPORTED_STREAM;
        if (rc != MD_ERR_UNSUPPORTED_STREAM) return rc;
        variant = 0;
    }
    const bool fast = a.Cin % 64 == 0 && a.kh * a.kw <= 32;  // MODE 2 preconditions (then Kpad == Kreal)
    // 3x3 / stride 1 / pad 1 with korder-1 weights: halo-reuse kernel (variant 0 auto or 11 forced)
    const bool halo_ok = dma_ok && !a.adv && a.korder == 1 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 &&
                         a.Cin % 64 == 0 && ctile == 128;
    // MFMA-bound layers (K >= 1024, Cout a multiple of 256): the 256x256 ping-pong kernel (measured r01, tools/conv_ab.py:
    // +17 % over the halo kernel on 3x3 256->256, +75 % on the 12544->1024 FC; loses on the HBM-bound K < 1024 layers)
    // and it needs enough 256x256 tiles to fill whole rounds of the 256 CUs (one workgroup per CU, ~1.2x the per-CU rate
    // of four resident 128x128 workgroups; a partial last round costs a full tile time there, a large fraction of a round
    // on the 128x128 kernel -- see below).  Unit: one 128x128 tile at the full-CU 128x128 rate.
    const long long pp_blocks = (M + 255) / 256 * (a.Cout / 256), sb_blocks = (M + 127) / 128 * ((a.Cout + 127) / 128);
    const double t_pp = (double)((pp_blocks + 255) / 256) * (4.0 / 1.2);
    const double sb_last = (double)(sb_blocks % 1024) / 1024.0;
    // a partial last round is expensive on the single-buffer kernel: the few workgroups left run alone on their CUs, nobody hides
    // their DMA latency (~1.3 us per K tile).  Measured r01, batch 32: 256->256 3x3 @50x84 = 2.05 rounds: 194 us vs ping-pong
    // (3 rounds) 176 us; 1024->256 1x1 @50x84: 107 vs 98 us -> a partial round costs 0.6-1.0 of a full one.
    const double sb_part = 2.5 + 1.5 * sb_last, sb_lone = 0.0726 * (double)(a.Kpad / BK);
    const double t_sb = (double)(sb_blocks / 1024) * 4.0 + (sb_blocks % 1024 ? (sb_part > sb_lone ? sb_part : sb_lone) : 0.0);
    const bool pp_ok = fast && dma_ok && a.Cout % 256 == 0 && a.Kpad >= 1024 && pp_blocks >= 128 && t_pp <= t_sb;
    // HALO form of the ping-pong kernel (3x3 / s1 / p1, korder-1 weights): 16 x 16-pixel tiles, the 18 x 18 halo staged once per channel
    // chunk instead of the B tile once per tap.  Auto where the 2-D tiles cover the image with little waste and the layer is long enough
    // for the energy per K tile to matter (r03 tools/pp_halo_ab.py)
    const long long halo_tiles = (long long)a.N * pingpong_halo_tiles_per_image(a.H, a.W, nullptr, nullptr);
    const bool halo_plain_out = !a.adv || (a.os == 1 && a.oy == 0 && a.ox == 0 && a.Ho == a.Hf && a.Wo == a.Wf);
    const bool halo_ok_pp = fast && dma_ok && pingpong_halo_takes(a) && halo_plain_out && a.Cout % 256 == 0;
    // r03 tools/pp_halo_ab.py (batch 60, same box, interleaved, bit-identical; with the conflict-free lane -> pixel map): on 200x336 (0.2 % idle
    // tile pixels) the fused-head form gains 10.3 %, the persistent form 3.4 %, the one-tile form 8.5 % (= the persistent HALO form); on
    // 100x168 (9 % idle) the head form +3.7 %, the persistent form -3 %; on 50x84 (22 % idle) everything loses 8-14 %.  In the step
    // (profiles/r03_pp_halo_step_ab.txt): head form -0.59 ms, persistent P2 form another -0.40 ms -> auto for the head form from 90 % tile
    // efficiency, for the persistent form from 99 %
    const double halo_eff = (double)M / (double)(halo_tiles * 256);
    const bool halo_auto = halo_ok_pp && !no_halo && ((head != nullptr && halo_eff >= 0.90) || (!halo_not_pers && halo_eff >= 0.99));
    const bool halo_pp = halo_ok_pp && (force_halo || halo_auto || (halo_fit && halo_eff >= 0.99));
    if (head) {
        if (!(fast && dma_ok && a.Cout == 256 && !a.adv && !a.res && a.relu == 1 && pp_blocks >= 64)) return MD_ERR_UNSUPPORTED_INTERNAL;
        a.w2 = head->w2; a.b2 = head->b2; a.y2 = head->y2;
        return launch_conv_pingpong_head(a, s, halo_pp);
    }
    // 3x3 layers on <= 256 channels: the 64-cout halo-reuse kernel at four workgroups per CU beats the 128x128 kernel wherever
    // the ping-pong kernel does not apply, and beats the ping-pong kernel when its 256x256 tiles fill the last of several rounds
    // badly (r01 tools/conv_ab_yolo.py, batch 32: 128->128 @80x80 +11 %, 256->256 @20x20 +16 %, 256->256 @80x80 (3.1 rounds) +8 %;
    // one-round grids and K = 4608 layers stay where they were -- except Cout not a multiple of 128, where the 128x128 kernel pads:
    // 512->320 @40x40 +23 %, tools/dispatch_audit.py)
    const long long pp_rounds = (pp_blocks + 255) / 256;
    const bool pp_ragged = pp_rounds >= 2 && (double)pp_blocks < 0.85 * (double)(pp_rounds * 256);
    // ... provided its 8x16-pixel tiles cover the image without much waste (100x168: 8 % idle lanes and the 128x128 kernel is 3 %
    // ahead), or the whole grid is resident at once anyway (<= 1024 workgroups: latency-bound, the halo kernel's shorter chain wins)
    const long long h_tiles = (long long)a.N * ((a.H + HT_H - 1) / HT_H) * ((a.W + HT_W - 1) / HT_W);
    const bool halo_fits = (double)a.H * a.W * a.N >= 0.95 * (double)(h_tiles * HT_H * HT_W) || h_tiles * (a.Cout / 64) <= 1024;
    const bool cat_only_h = !a.adv || (a.os == 1 && a.oy == 0 && a.ox == 0 && a.Ho == a.Hf && a.Wo == a.Wf && a.pad_top == a.pad &&
                                       a.pad_left == a.pad && (!a.res || a.Rs));
    const bool halo64_first = variant == 0 && !head && dma_ok && cat_only_h && a.korder == 1 && a.kh == 3 && a.kw == 3 && a.stride == 1 &&
                              a.pad == 1 && a.Cin % 64 == 0 && (a.Cin <= 256 || (a.Cin <= 512 && a.Cout % 128 != 0)) && a.Cout % 64 == 0 &&
                              cout_pad % 64 == 0 && !a.res_up &&
                              a.Ho == a.H && a.Wo == a.W && (!pp_ok || pp_ragged) && halo_fits;
    if (halo64_first) return launch_conv3x3_halo<64, true>(a, s);
    // the persistent form of the ping-pong kernel (no residual, plain / concat output, the cout tiles divide a 32-CU XCD, more than one
    // round of tiles): auto for the long-K layers (r02 tools/pp_pers_ab.py, batch 60, bit-identical: 3x3 256->256 +3.2...4.4 %, 3x3 512->512
    // +4.2 %; the K = 1024 1x1 layers lose 4-5 %: behind a tile's prologue DMAs sit the previous tile's stores, and the loop's first counted
    // wait then covers their write latency -- 1 / 16 of such a tile's K loop); variant 32 forces it, 33 = auto without it
    {
        const bool cat_only_p = !a.adv || (a.os == 1 && a.oy == 0 && a.ox == 0 && a.Ho == a.Hf && a.Wo == a.Wf);
        const bool pers_ok = fast && dma_ok && a.Cout % 256 == 0 && !a.res && !a.res_up && cat_only_p && 32 % (a.Cout / 256) == 0 && pp_blocks > 256;
        const bool pers_auto = variant == 0 && !no_pers && pp_ok && pers_ok && a.Kpad >= tn.pers_min_k &&
                               !(halo_not_pers && halo_ok_pp && (double)M >= 0.99 * (double)(halo_tiles * 256));
        if ((variant == 32 || pers_auto) && pers_ok) {
            // (persistent + HALO exists on the 16x16x32 MFMA shape only: the 32x32x16 instantiation needs 258 registers)
            if (halo_pp && 32 % (a.Cout / 256) == 0 && halo_tiles * (a.Cout / 256) > 256) return launch_conv_pingpong_pers<1, true>(a, s);
            return pingpong_wants_16x16(a) ? launch_conv_pingpong_pers<1>(a, s) : launch_conv_pingpong_pers<0>(a, s);
        }
        if (variant == 32) variant = 0;
    }
    if (variant == 0 && pp_ok && (halo_pp || (halo_not_pers && halo_ok_pp && (double)M >= 0.99 * (double)(halo_tiles * 256)))) return pingpong_wants_16x16(a) ? launch_conv_pingpong_halo<1>(a, s) : launch_conv_pingpong_halo<0>(a, s);
    if (variant == 0 && pp_ok) return pingpong_wants_16x16(a) ? launch_conv_pingpong<0, 1>(a, s) : launch_conv_pingpong<0>(a, s);
    if (halo_ok && !a.res_up && variant == 11) return launch_conv3x3_halo<128, false>(a, s);  // superseded by the paths around it
    // 64-cout tiles of the halo kernel at four workgroups per CU (variant 27; auto for Cout <= 64)
    // (a channel-concat output is fine: the kernel stores with the output tensor's channel stride; sub-pixel addressing is not)
    const bool cat_only = !a.adv || (a.os == 1 && a.oy == 0 && a.ox == 0 && a.Ho == a.Hf && a.Wo == a.Wf && a.pad_top == a.pad &&
                                     a.pad_left == a.pad && (!a.res || a.Rs));
    const bool halo64_ok = dma_ok && cat_only && a.korder == 1 && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 &&
                           a.Cin % 64 == 0 && a.Cout % 64 == 0 && cout_pad % 64 == 0 && !a.res_up && a.Ho == a.H && a.Wo == a.W;
    if (halo64_ok && (variant == 27 || (variant == 0 && ctile == 64))) return launch_conv3x3_halo<64, true>(a, s);
    if (variant == 11) variant = 2;
    // 36 / 37: the HALO form pinned (32x32x16 / 16x16x32 MFMA), 38: its persistent form; where it does not apply: the plain ping-pong kernel
    if ((variant == 36 || variant == 37 || variant == 38) && fast && dma_ok && a.Cout % 256 == 0) {
        const bool gen_plain = (!a.adv || halo_plain_out) && !a.res_up;
        if (halo_ok_pp && gen_plain) {
            if (variant == 38 && !a.res && 32 % (a.Cout / 256) == 0 && halo_tiles * (a.Cout / 256) > 256) return launch_conv_pingpong_pers<1, true>(a, s);
            return variant == 36 ? launch_conv_pingpong_halo<0>(a, s) : launch_conv_pingpong_halo<1>(a, s);
        }
        return variant == 36 ? launch_conv_pingpong<0>(a, s) : launch_conv_pingpong<0, 1>(a, s);
    }
    if (variant == 15 && fast && dma_ok && a.Cout % 256 == 0) return launch_conv_pingpong<0>(a, s);  // 256x256 ping-pong, 8 waves
    if (variant == 22 && fast && dma_ok && a.Cout % 256 == 0) return launch_conv_pingpong<0, 1>(a, s);  // same, 16x16x32 MFMA
#ifdef MD_DIAG
    if (variant == 26 && fast && dma_ok && a.Cout % 256 == 0) return launch_conv_pingpong<4, 1>(a, s);   // stamps, 16x16x32 MFMA
    if (variant >= 17 && variant <= 19 && fast && dma_ok && a.Cout % 256 == 0)                       // timing ablations
        return variant == 17 ? launch_conv_pingpong<1>(a, s) : (variant == 18 ? launch_conv_pingpong<2>(a, s) : launch_conv_pingpong<4>(a, s));
#endif
    if (ctile != 128) {
        if (variant == 1) return ctile == 64 ? launch_conv<256, 1, 4, 2, 2, 0>(a, s) : launch_conv<256, 1, 4, 1, 2, 0>(a, s);
        if (fast) return ctile == 64 ? launch_conv<256, 1, 4, 2, 2, 2>(a, s) : launch_conv<256, 1, 4, 1, 2, 2>(a, s);
        return ctile == 64 ? launch_conv<256, 1, 4, 2, 2, 1>(a, s) : launch_conv<256, 1, 4, 1, 2, 1>(a, s);
    }
    // a grid of fewer than two workgroups per CU with a long K loop: nobody else hides the DMA latency, stage the next tile
    // while this one is multiplied (512->512 @20x20, batch 32: +11 %)
    if (variant == 0 && sb_blocks <= 512 && a.Kpad / BK >= 4) a.single_buf = 0;
    if (variant == 1) return launch_conv<256, 2, 2, 2, 2, 0>(a, s);               // register-staged, 64-bit addressing
    return fast ? launch_conv<256, 2, 2, 2, 2, 2>(a, s) : launch_conv<256, 2, 2, 2, 2, 1>(a, s);
}

extern "C" int md_conv2d(MD_AOT_ARGS) { return conv2d_entry(nparam, params, ndims, shapes, dtypes, stream, extra, nullptr); }

// conv (Cout = 256, ReLU) followed by a 1x1 head with <= 16 output channels, in one launch where the ping-pong kernel
// applies (the 256-channel intermediate then never leaves the CU); otherwise the two convolutions run back to back
// through a stream-ordered temporary (or the caller's workspace).  The RPN head of the two-stage detectors: 3x3 conv +
// ReLU -> [objectness | deltas].
extern "C" int md_conv2d_head(MD_AOT_ARGS) {
    // in: x[N,H,W,Cin], w[256,Kpad], bias[256], w2[32,256] (rows >= c2 zero), bias2[32] ; out: y2[N,Ho,Wo,16] ; [workspace]
    if (nparam != 6 && nparam != 7) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || !params[3] || !params[4]) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 3, "bfloat16") || !dtype_is(dtypes, 4, "float32") || !dtype_is(dtypes, 5, "bfloat16")) return MD_ERR_ARG;
    if (ndims[0] != 4 || ndims[1] != 2 || ndims[3] != 2 || ndims[5] != 4) return MD_ERR_ARG;
    if (shapes[1][0] != 256 || shapes[3][0] < 16 || shapes[3][1] != 256 || numel(ndims, shapes, 4) < 16 || shapes[5][3] != 16 ||
        shapes[5][0] != shapes[0][0])
        return MD_ERR_ARG;
    const md_conv2d_attrs *at = (const md_conv2d_attrs *)extra;
    if (at->adv || at->relu != 1 || at->res_upsample) return MD_ERR_ARG;
    const int64_t N = shapes[5][0], Ho = shapes[5][1], Wo = shapes[5][2];
    if (N * Ho * Wo == 0) return MD_OK;
    if (!params[0] || !params[5]) return MD_ERR_ARG;
    // the conv as md_conv2d sees it: residual NULL, output [N,Ho,Wo,256] (never written by the fused kernel)
    int64_t sy[4] = {N, Ho, Wo, 256}, snull[1] = {0};
    int nd5[5] = {ndims[0], ndims[1], ndims[2], 0, 4};
    int64_t *sh5[5] = {shapes[0], shapes[1], shapes[2], snull, sy};
    const char *dt5[5] = {dtypes[0], dtypes[1], dtypes[2], nullptr, "bfloat16"};
    HeadArgs head = {(const uint16_t *)params[3], (const float *)params[4], (uint16_t *)params[5]};
    void *p5[5] = {params[0], params[1], params[2], nullptr, params[5] /* placeholder, not written */};
    int rc = conv2d_entry(5, p5, nd5, sh5, dt5, stream, extra, &head);
    if (rc != MD_ERR_UNSUPPORTED_INTERNAL) return rc;
    // two launches through a temporary [N,Ho,Wo,256]
    Scratch tmp;
    const size_t bytes = (size_t)(N * Ho * Wo) * 256 * 2;
    rc = tmp.acquire(bytes, nparam, params, ndims, shapes, 6, (hipStream_t)stream);
    if (rc != MD_OK) return rc;
    p5[4] = tmp.ptr;
    rc = conv2d_entry(5, p5, nd5, sh5, dt5, stream, extra, nullptr);
    if (rc != MD_OK) return rc;
    md_conv2d_attrs a1 = {};
    a1.kh = a1.kw = 1; a1.stride = 1; a1.pad = 0; a1.relu = 0; a1.variant = at->variant < 15 ? at->variant : 0;
    a1.tune = at->tune;
    int64_t sw2[2] = {shapes[3][0], 256}, sb2[1] = {shapes[3][0]};
    int nd1[5] = {4, 2, 1, 0, 4};
    int64_t *sh1[5] = {sy, sw2, sb2, snull, shapes[5]};
    const char *dt1[5] = {"bfloat16", "bfloat16", "float32", nullptr, "bfloat16"};
    void *p1[5] = {tmp.ptr, params[3], params[4], nullptr, params[5]};
    return conv2d_entry(5, p1, nd1, sh1, dt1, stream, &a1, nullptr);
}

// y = act(W . [x_a ; x_b sampled with stride] + bias [+ residual]): ONE 1x1 GEMM over the K-concatenation of two inputs.
// The use: the first block of a ResNet stage (centernet/src/resnet.py:139-178 with `downsample`, built by _make_layer :214-224):
// out = relu(bn3(conv3(t2)) + bn_d(conv_d(x))) -- conv3 is 1x1 on the block's own feature map, conv_d a 1x1 conv with the block's stride
// on its input.  [W3 | Wd] . [t2 ; x_strided] + (b3 + bd) is the same sum in one accumulator: no downsample launch, no Cout-channel
// residual tensor written and re-read.  (One bf16 rounding instead of the three of the layer-by-layer path.)
// in : x_a[N,Ho,Wo,Ca] bf16 (Ca % 64 == 0), x_b[N,Hb,Wb,Cb] bf16 (Cb % 64 == 0, Ho == (Hb-1)/stride_b + 1), w[Cout_pad, Ca + Cb] bf16,
//      bias[Cout_pad] f32, residual[N,Ho,Wo,Cout] bf16 | NULL ; out y[N,Ho,Wo,Cout] bf16 (Cout > 64).  extra: md_conv1x1_dual_attrs
extern "C" int md_conv1x1_dual(MD_AOT_ARGS) {
    if (nparam != 6) return MD_ERR_NPARAM;
    if (!params || !extra || !ndims || !shapes || !params[2] || !params[3]) return MD_ERR_ARG;
    if (!dtype_is(dtypes, 0, "bfloat16") || !dtype_is(dtypes, 1, "bfloat16") || !dtype_is(dtypes, 2, "bfloat16") ||
        !dtype_is(dtypes, 3, "float32") || !dtype_is(dtypes, 5, "bfloat16") || (params[4] && !dtype_is(dtypes, 4, "bfloat16")))
        return MD_ERR_ARG;
    if (ndims[0] != 4 || ndims[1] != 4 || ndims[2] != 2 || ndims[5] != 4) return MD_ERR_ARG;
    const md_conv1x1_dual_attrs *at = (const md_conv1x1_dual_attrs *)extra;
    const int64_t N = shapes[0][0], Ho = shapes[0][1], Wo = shapes[0][2], Ca = shapes[0][3];
    const int64_t Hb = shapes[1][1], Wb = shapes[1][2], Cb = shapes[1][3], Cout = shapes[5][3];
    if (at->stride_b < 1 || at->relu < 0 || at->relu > 1 || Ca % 64 || Cb % 64 || Ca < 64 || Cb < 64 || Cout % 8 || Cout <= 64) return MD_ERR_ARG;
    if (shapes[1][0] != N || shapes[5][0] != N || shapes[5][1] != Ho || shapes[5][2] != Wo || Hb < 1 || Wb < 1 ||
        Ho != (Hb - 1) / at->stride_b + 1 || Wo != (Wb - 1) / at->stride_b + 1)
        return MD_ERR_ARG;
    const int64_t cout_pad = (Cout + 127) / 128 * 128;
    if (shapes[2][0] != cout_pad || shapes[2][1] != Ca + Cb || numel(ndims, shapes, 3) != cout_pad) return MD_ERR_ARG;
    if (params[4] && (ndims[4] != 4 || numel(ndims, shapes, 4) != N * Ho * Wo * Cout)) return MD_ERR_ARG;
    if (N * Ho * Wo == 0) return MD_OK;
    if (!params[0] || !params[1] || !params[5]) return MD_ERR_ARG;
    if (Hb > 32000 || Wb > 32000) return MD_ERR_SIZE;
    const long long xa_img = Ho * Wo * Ca * 2, xb_img = Hb * Wb * Cb * 2, w_bytes = cout_pad * (Ca + Cb) * 2;
    const long long big = xa_img > xb_img ? xa_img : xb_img;
    if (big >= 0x7fff0000LL || w_bytes >= 0x7fff0000LL) return MD_ERR_SIZE;
    const Tune tn = resolve_tune(&at->tune);
    const long long lim = tn.chunk_limit > big ? tn.chunk_limit : big;
    const long long per = lim / big < N ? lim / big : N;     // images per launch: both inputs stay inside the DMA reach
    for (long long n0 = 0; n0 < N; n0 += per) {
        const long long nn = N - n0 < per ? N - n0 : per;
        ConvArgs a = {};
        a.x = (const uint16_t *)params[0] + n0 * Ho * Wo * Ca;
        a.x2 = (const uint16_t *)params[1] + n0 * Hb * Wb * Cb;
        a.w = (const uint16_t *)params[2]; a.bias = (const float *)params[3];
        a.res = params[4] ? (const uint16_t *)params[4] + n0 * Ho * Wo * Cout : nullptr;
        a.y = (uint16_t *)params[5] + n0 * Ho * Wo * Cout;
        a.N = (int)nn; a.H = (int)Ho; a.W = (int)Wo; a.Cin = (int)Ca; a.Xs = (int)Ca; a.Cout = (int)Cout; a.Ho = (int)Ho; a.Wo = (int)Wo;
        a.kh = a.kw = 1; a.stride = 1; a.pad = 0; a.relu = at->relu;
        a.Kpad = (int)(Ca + Cb); a.Kreal = a.Kpad; a.cpt = (int)(Ca / 8);
        a.M = (int)(nn * Ho * Wo);
        a.x_bytes = (unsigned)(nn * xa_img); a.w_bytes = (unsigned)w_bytes; a.x2_bytes = (unsigned)(nn * xb_img);
        a.Hf = (int)Ho; a.Wf = (int)Wo; a.Ctot = (int)Cout; a.os = 1; a.pointwise = 1;
        a.H2 = (int)Hb; a.W2 = (int)Wb; a.Xs2 = (int)Cb; a.stride2 = at->stride_b; a.nk_a = (int)(Ca / 64);
        const int rc = launch_conv_dual(a, (hipStream_t)stream, tn);
        if (rc != MD_OK) return rc;
    }
    return MD_OK;
}
