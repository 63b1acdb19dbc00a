/*
 * minddet_hip.h -- C ABI of libminddet_hip.so (hand-written HIP for gfx950 / MI355X).
 *
 * Every entry point uses the MindSpore "AOT custom operator" convention that the reference
 * uses for its own native ops, so the library is a drop-in behind ops.Custom(func_type="aot"):
 *
 *   reference CPU op : extern "C" int boxes_iou_nms_cpu(int nparam, void **params, int *ndims,
 *                      int **shapes, const char **dtypes, void *stream, void *extra)
 *                      minddet/models/centerpoint/det3d_ms/ops/iou-bev-nms-org.cpp:237
 *   reference GPU ops: extern "C" int NmsGpu(int nparam, void** params, int* ndims,
 *                      int64_t** shapes, const char** dtypes, void* stream, void* extra)
 *                      minddet/models/centerpoint/det3d_ms/ops/test_custom_pytorch/
 *                      iou3d_nms_kernel.cu:445,467,491,548
 *
 * Conventions (all ops):
 *   - params[] = inputs, then outputs, in the order documented per op; an OPTIONAL trailing
 *     workspace buffer (dtype "uint8") may follow the outputs -- when absent the op uses the
 *     library's scratch pool: ONE buffer per (device, stream), grown on demand and reused by
 *     every later call on that stream (no driver call in the steady state; md_scratch_release
 *     frees it).
 *   - every pointer in params[] is DEVICE memory owned by the caller; outputs are fixed
 *     shape and padded (keep[N] has `num` valid leading entries, the rest 0 --
 *     iou-bev-nms-org.cpp:247-249,274-281).
 *   - ndims[i] / shapes[i][j] describe params[i]; dtypes[i] is one of "uint8","int8","int16",
 *     "int32","int64","float16","bfloat16","float32","float64" (ms_ext.cpp:6-7 + bfloat16).
 *   - stream is a hipStream_t (may be NULL = default stream).  Work is ENQUEUED on it; the
 *     ops never synchronise the device (the reference's cudaStreamSynchronize +
 *     cudaMemcpy round trip, iou3d_nms_kernel.cu:448-449,515-517, is what this removes).
 *   - `extra`: NULL, or a pointer to the op's attribute struct declared below (host memory,
 *     read during the call only).
 *   - return 0 on success; non-zero on failure (iou-bev-nms-org.cpp:238,282): 1 = wrong
 *     nparam, 2 = bad dtype/shape, 3 = HIP runtime error, 4 = unsupported size.  No
 *     exceptions, no stdout, no exit().
 *   - re-entrant; safe to call from several host threads on different streams.  Process-lifetime
 *     data: per-thread diagnostics counters (md_conv2d_last_kernel / md_conv2d_launch_count), a
 *     "LDS size attribute set" cache per kernel and device, and the scratch pool above -- each
 *     behind its own lock.  No tuning state: every knob is a PER-CALL attribute (md_conv_tune).
 */
#ifndef MINDDET_HIP_H_
#define MINDDET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MD_AOT_ARGS                                                                           \
    int nparam, void **params, int *ndims, int64_t **shapes, const char **dtypes, void *stream, \
        void *extra

#define MD_OK 0
#define MD_ERR_NPARAM 1
#define MD_ERR_ARG 2
#define MD_ERR_HIP 3
#define MD_ERR_SIZE 4

/* library / build info: returns a static string "minddet_hip <ver> gfx950". */
const char *md_version(void);
/* frees the scratch pool (hipFree: waits for the device).  Optional: call before unloading the library or to give the memory
 * back; ops called afterwards grow a new pool.  Returns MD_OK / MD_ERR_HIP.  (Under stream capture pass the workspace param.) */
int md_scratch_release(void);

/* ------------------------------------------------------------------------------------------
 * Rotated (BEV) IoU + NMS -- same symbol names and parameter lists as the reference's AOT GPU
 * ops (test_custom_pytorch/iou_gpu.py:14-81), so `"./iou_nms.so:NmsGpu"` can simply be
 * re-pointed at this library.
 * ------------------------------------------------------------------------------------------ */
/* in: boxes_a[M,7] f32, boxes_b[N,7] f32 ; out: iou[M,N] f32.  iou3d_nms_kernel.cu:251-265 */
int BoxesIouBevGpu(MD_AOT_ARGS);
/* in: boxes_a[M,7] f32, boxes_b[N,7] f32 ; out: overlap[M,N] f32.  iou3d_nms_kernel.cu:236-249 */
int BoxesOverlapBevGpu(MD_AOT_ARGS);
/* in: boxes[N,7] f32 (sorted by score, descending), thresh[1] f32 ;
 * out: keep[N] i64 (leading num valid, rest 0), num[1] i32.  Suppress iff IoU > thresh,
 * IoU = overlap / fmaxf(sa+sb-overlap, 1e-8).  iou3d_nms_kernel.cu:267-311 + :491-546 */
int NmsGpu(MD_AOT_ARGS);
/* same I/O; axis-aligned IoU of the (x,y,dx,dy) footprint.  iou3d_nms_kernel.cu:314-372,548-601 */
int NmsNormalGpu(MD_AOT_ARGS);
/* Device twin of the reference's CPU operator boxes_iou_nms_cpu (iou-bev-nms-org.cpp:237-283,
 * Python side nms_cpu.py:10-27): in: boxes[N,7] f32, thresh[1] f32 ; out: keep[N] i32,
 * num[1] i32.  Suppress iff ovr >= thresh with ovr = overlap/(sa+sb-overlap) (no eps);
 * zero-area boxes are dropped up front.  N comes from shapes[0][0] (the reference hard-codes
 * 1000, :244). */
int boxes_iou_nms_gpu(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * Axis-aligned IoU / NMS family
 * ------------------------------------------------------------------------------------------ */
typedef struct md_iou_attrs {
    float eps; /* added to every width/height; pointpillars/src/core/box_np_ops.py:639 */
} md_iou_attrs;
/* iou_jit: in boxes[N,4] f32, query[K,4] f32 ; out overlaps[N,K] f32.  extra: md_iou_attrs
 * (NULL -> eps 0).  pointpillars/src/core/box_np_ops.py:639-679 */
int md_iou_aligned(MD_AOT_ARGS);

typedef struct md_rotate_iou_attrs {
    int32_t criterion; /* -1 IoU, 0 inter/area(box), 1 inter/area(query), 2 raw intersection */
} md_rotate_iou_attrs;
/* rotate_iou_kernel_eval (pointpillars/eval_gpu/rotate_iou.py:264-340): in boxes[N,5] f32 (cx,cy,dx,dy,angle),
 * query[K,5] f32 ; out iou[N,K] f32.  The reference kernel runs under numba.cuda and has no fixture: parity unpinned. */
int md_rotate_iou_eval(MD_AOT_ARGS);

typedef struct md_nms_attrs {
    float iou_threshold;
    float eps;         /* mode 0 only */
    int32_t mode;      /* 0 nms_jit (>= thr, eps)       pointpillars/src/core/nms.py:85-112
                          1 apply_nms (+1 px, > thr)     pointpillars/src/core/nms.py:7-41
                          2 strict > thr, fmaxf(union,1e-8) (iou_normal on corner boxes) */
    int32_t max_output; /* <=0: no cap; else keep only the first max_output survivors */
} md_nms_attrs;
/* Greedy NMS over B independent lists of corner boxes already sorted by descending score.
 * in : boxes[B,N,4] f32 (or [N,4]), count[B] i32 (valid leading boxes per list; may be a
 *      NULL pointer = all N), group[B,N] i32 (class / task key: boxes with different keys
 *      never suppress each other; may be a NULL pointer)
 * out: keep_mask[B,N] u8, keep_idx[B,N] i32 (leading num valid, rest 0), num[B] i32
 * extra: md_nms_attrs (required). */
int md_nms_aligned(MD_AOT_ARGS);

typedef struct md_soft_nms_attrs {
    float sigma, Nt, threshold; /* reference call: sigma 0.5 (default), Nt 0.5, threshold 0.001 */
    int32_t method;             /* 1 linear, 2 gaussian, 3 hard */
} md_soft_nms_attrs;
/* Soft-NMS (Bodla et al. 2017; call site centernet/src/post_process.py:45-52 -- the Cython module is not vendored in
 * the reference, parity unpinned).  in boxes[L,N,4] f32, scores[L,N] f32, count[L] i32 or NULL (N <= 1024) ;
 * out scores_out[L,N] f32 (decayed score of each surviving box at its ORIGINAL position, 0 = removed),
 * order[L,N] i32 (survivors in selection order, leading num valid), num[L] i32 */
int md_soft_nms(MD_AOT_ARGS);

/* circle_nms (centerpoint/det3d_ms/core/utils/circle_nms_jit.py:6-36):
 * in xy[N,2] f32 sorted by score desc, thresh[1] f32 ; out keep_mask[N] u8, keep_idx[N] i32,
 * num[1] i32.  Suppress iff squared centre distance <= thresh. */
int md_circle_nms(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * Conv + folded BN + (residual) + ReLU, implicit GEMM on MFMA (bf16 in, fp32 accumulate)
 * ------------------------------------------------------------------------------------------ */
/* Per-call tuning knobs of the conv family (A/B tools and the tests that drive the chunked / multi-round paths on small tensors).
 * All zero = the library's defaults; nothing persists between calls and nothing is shared between threads. */
typedef struct md_conv_tune {
    int32_t chunk_limit;       /* activation bytes above which the op runs the batch as consecutive image chunks on the same stream
                                  (the kernels' 32-bit LDS-DMA offsets); 0 = the default and maximum, 2 GiB - 64 KiB */
    int32_t stream_rounds;     /* conv1x1_stream_kernel: workgroup rounds the pixel range is cut into; 0 = default 1 */
    int32_t stream_wgs_per_cu; /* conv1x1_stream_kernel: workgroups per CU the grid is sized for; 0 = default 2 */
    int32_t stream_cache_bits; /* conv1x1_stream_kernel cache policy: 0 = default (6); else 8 | bits (1 = activation DMA nt,
                                  2 = residual DMA nt, 4 = stores nt); | 16 = the K = 512 form on 128-cout (4-wave) workgroups also
                                  where the 256-cout (8-wave) form applies (A/B) */
    int32_t pers_min_k;        /* the persistent form of the ping-pong kernel is the dispatcher's choice for eligible layers with
                                  K >= this; 0 = default 2304 */
    int32_t dual_pp_min_k;     /* md_conv1x1_dual runs on the ping-pong kernel when its concatenated K is >= this (and Cout % 256 == 0,
                                  no residual tensor); 0 = default 768 */
} md_conv_tune;

typedef struct md_conv2d_attrs {
    int32_t kh, kw, stride, pad; /* square stride / symmetric zero padding */
    int32_t relu;                /* activation: 0 none; 1 ReLU applied after bias (+ residual);
                                    2 SiLU applied after bias, BEFORE the residual add (x + act(conv(x))) */
    int32_t variant;             /* 0 = auto (default: cost model in csrc/conv.hip).  Pins a kernel for A/B measurements:
                                    1 register-staged 128x128, 2 / 20 LDS-DMA 128x128 with two / one staging buffer,
                                    11 / 27 halo-reuse kernel with 128- / 64-cout tiles,
                                    15 / 22 256x256 ping-pong kernel (32x32x16 / 16x16x32 MFMA), 32 its persistent form, 36 / 37 / 38 its
                                    HALO form for 3x3 layers (16x16-pixel tiles, the halo staged once per channel chunk; 38 persistent);
                                    30 weight-stationary pointwise kernel; 31 / 33 / 35 = auto without the pointwise / persistent /
                                    HALO kernel, 34 = auto with the HALO form wherever it applies, 39 / 40 = auto with the HALO form also for / instead of
                                    the persistent form where its tiles fit (A/B records in profiles/r03_pp_halo_step_ab.txt); a variant whose
                                    preconditions do not hold falls back to the generic kernel.  17-19 and 25 (timing /
                                    stamp diagnostics that do NOT compute the convolution) exist only in the MD_DIAG build
                                    used by tools/ (libminddet_hip_diag.so); this library rejects them with MD_ERR_ARG. */
    /* generalised addressing, used when adv != 0 (all zero = plain conv).  The op then computes, for
     * ho < sub_h, wo < sub_w:  y[n, ho*out_stride + out_off_y, wo*out_stride + out_off_x, c_off + c] =
     * act(bias[c] + sum x[n, ho*stride - pad_top + kh, wo*stride - pad_left + kw, ci] * w[c,kh,kw,ci]), c < cout.
     * This is how transposed convs (one launch per output-pixel parity, Conv2dTranspose k=4 s=2 p=1 of
     * centernet/src/centernet_det.py:145-152 and k=s of centerpoint/det3d_ms/models/necks/rpn.py:66-80)
     * and channel-concatenated outputs (rpn.py:152, pointpillars.py:598) run on the same MFMA kernel. */
    int32_t adv;
    int32_t pad_top, pad_left, sub_h, sub_w, out_stride, out_off_y, out_off_x, c_off, cout;
    int32_t res_upsample;        /* 1: `residual` is [N, ceil(Ho/2), ceil(Wo/2), Cout] and is added with nearest 2x
                                    upsampling (FPN top-down add fused into the lateral conv); plain addressing only */
    int32_t korder;              /* K order of the packed weights: 0 = (kh,kw,ci) [default]; 1 = (ci/64, kh,kw, ci%64)
                                    (needs Cin % 64 == 0): taps innermost, so consecutive K tiles of a 3x3 window
                                    re-read nearly the same activation lines */
    /* channel-slice operands (all zero = whole tensors), so that split / concat graphs (C2f, C3: chunk -> bottlenecks ->
     * concat) need no copies: every branch reads and writes its channel range of ONE concat buffer. */
    int32_t x_c_off, x_cin;      /* x_cin > 0: the conv reads channels [x_c_off, x_c_off + x_cin) of x[N,H,W,C] (both % 8 == 0;
                                    the packed weights are for Cin = x_cin) */
    int32_t res_slice, res_c_off;/* res_slice != 0: residual is [N,Ho,Wo,R] and channels [res_c_off, res_c_off + Cout) are added
                                    (res_c_off % 8 == 0; unit output stride, no res_upsample) */
    int32_t reserved0;           /* must be 0 (MD_ERR_ARG otherwise) */
    md_conv_tune tune;           /* all zero = defaults */
} md_conv2d_attrs;
/* Replaces Conv2d -> BatchNorm2d(eval) -> [+ residual] -> ReLU of the reference graphs
 * (centernet/src/resnet.py:109-178,181-252; centerpoint/det3d_ms/models/necks/rpn.py:9-154).
 * in : x[N,H,W,Cin] bf16 NHWC (Cin % 8 == 0),
 *      w[Cout_pad, Kpad] bf16, K ordered (kh,kw,ci), BN folded (w' = w*gamma/sqrt(var+eps)),
 *        Kpad = roundup(kh*kw*Cin, 64) zero padded, Cout_pad = roundup(Cout, md_conv2d_cout_tile(Cout)),
 *      bias[Cout_pad] f32 (b' = beta - mean*gamma/sqrt(var+eps)),
 *      residual[N,Ho,Wo,Cout] bf16 or a NULL pointer
 * out: y[N,Ho,Wo,Cout] bf16 (Cout % 8 == 0).   extra: md_conv2d_attrs (required). */
int md_conv2d(MD_AOT_ARGS);
/* tile of output channels the dispatcher uses for a given Cout (32, 64 or 128): the packer
 * pads Cout up to a multiple of it.  Pure function, callable without a GPU. */
int md_conv2d_cout_tile(int cout);

/* ONE 1x1 GEMM over the K-concatenation of two inputs: y = act(W . [x_a ; x_b sampled with stride_b] + bias [+ residual]).
 * Replaces, in the first block of a ResNet stage (centernet/src/resnet.py:139-178 with `downsample`; _make_layer :214-224),
 * conv3 + bn3 on the block's feature map AND the strided 1x1 downsample conv + bn on the block input AND their add:
 * w = [w3 | wd] (both BN-folded), bias = b3 + bd.  The Cout-channel residual tensor of the layer-by-layer form is never written
 * (-2 x Cout x 2 B per pixel of HBM traffic, one launch less); the sum is rounded to bf16 once instead of three times.
 * in : x_a[N,Ho,Wo,Ca] bf16, x_b[N,Hb,Wb,Cb] bf16 (Ca, Cb multiples of 64; Ho = (Hb-1)/stride_b + 1), w[roundup(Cout,128), Ca+Cb] bf16,
 *      bias[roundup(Cout,128)] f32, residual[N,Ho,Wo,Cout] bf16 | NULL ; out y[N,Ho,Wo,Cout] bf16 (Cout > 64). */
typedef struct md_conv1x1_dual_attrs {
    int32_t stride_b;   /* spatial stride of x_b (the block's stride) */
    int32_t relu;       /* 0 none, 1 ReLU after the sum */
    md_conv_tune tune;  /* all zero = defaults (chunk_limit, dual_pp_min_k are read) */
} md_conv1x1_dual_attrs;
int md_conv1x1_dual(MD_AOT_ARGS);

/* Conv (Cout = 256) + bias + ReLU followed by a 1x1 conv with <= 16 output channels (the RPN head of the two-stage
 * detectors: shared 3x3 conv -> [objectness | deltas]); where the 256x256 ping-pong kernel applies, the 256-channel
 * intermediate never leaves the CU (second GEMM straight from the epilogue image), otherwise two launches through a
 * temporary.  Absent from the reference (its Faster R-CNN is a README bullet): parity unpinned.
 * in : x[N,H,W,Cin] bf16, w[256,Kpad] bf16, bias[256] f32, w2[32,256] bf16 (rows >= 16 ignored), bias2[32] f32
 * out: y2[N,Ho,Wo,16] bf16 ; optional trailing workspace (N*Ho*Wo*512 bytes, used only by the two-launch path)
 * extra: md_conv2d_attrs of the FIRST conv (relu must be 1, plain addressing). */
int md_conv2d_head(MD_AOT_ARGS);

/* A whole ResNet bottleneck block (centernet/src/resnet.py:139-178: conv1 1x1 + bn + relu -> conv2 3x3 + bn + relu -> conv3 1x1 +
 * bn, + residual, relu) with 64 mid channels, stride 1 and 256 output channels in ONE launch: both 64-channel intermediates stay
 * in LDS and x is read from HBM once (1024 B of HBM traffic per pixel instead of 2048 B for the three md_conv2d launches, which are
 * HBM-bound).  Same arithmetic per layer as md_conv2d (bf16 operands, fp32 accumulation in the same K order, bf16-rounded
 * intermediates), so the result equals the three-launch path bit for bit.
 * in : x[N,H,W,Cin] bf16 (Cin = 64 or 256), w1[64,Cin] bf16, b12[128] f32 (conv1's 64 biases followed by conv2's),
 *      w2[64,576] bf16 (K = tap*64 + ci), w3[256,64] bf16, b3[256] f32 (weights as md_conv2d packs them, BN folded),
 *      residual[N,H,W,256] bf16 | NULL, wd[256,64] bf16 | NULL, bd[256] f32 | NULL -- the residual source is, in this order:
 *        wd / bd given (Cin == 64, residual NULL): the block's 1x1 downsample conv bf16(wd . x + bd), computed in the same launch
 *          from the x tile in LDS (the first block of a stage, resnet.py _make_layer: no downsample launch, no 512 B / pixel tensor);
 *        residual given: that tensor;   neither (Cin == 256): x itself (identity block).
 * out: y[N,H,W,256] bf16.   extra: NULL or md_conv_tune (chunk_limit is read). */
int md_bottleneck(MD_AOT_ARGS);

/* The Bottleneck of a YOLOv5 C3 block (build-authored model of BASELINE configs[1]; the reference names the family only,
 * README.md:5-14) in ONE launch:  y[.., y_c_off : +C] = x[.., x_c_off : +C] (if shortcut) + silu(conv3x3(silu(conv1x1(x[.., x_c_off : +C])))),
 * C = 64 or 128 channels in and out, BN folded, stride 1 / pad 1; the C-channel intermediate stays in LDS.  Same arithmetic as the two
 * md_conv2d launches it replaces (bf16 operands, fp32 accumulation in the same K order, the intermediate and the pre-shortcut value
 * rounded to bf16): bit-identical to them.  With pass_through the next C channels of x are copied to the next C channels of y (the
 * cv2(x) half of the C3 concat buffer travels with the result).  x and y must not overlap: MD_ERR_ARG.
 * in : x[N,H,W,XC] bf16, w1[C,C] bf16, b12[2C] f32 (conv1's biases, then conv2's), w2[C,9C] bf16 (md_conv2d's korder-1 layout:
 *      K = (ci / 64) * 576 + tap * 64 + ci % 64);  out: y[N,H,W,YC] bf16.   extra: md_c3_pair_attrs (required). */
typedef struct md_c3_pair_attrs {
    int32_t x_c_off;       /* first channel of x the pair reads (multiple of 8) */
    int32_t y_c_off;       /* first channel of y it writes (multiple of 8) */
    int32_t shortcut;      /* 1: add x (YOLOv5 Bottleneck(shortcut=True)) */
    int32_t pass_through;  /* 1: also copy x[.., x_c_off + C : + 2C] -> y[.., y_c_off + C : + 2C] */
} md_c3_pair_attrs;
int md_c3_pair(MD_AOT_ARGS);

/* Which kernel the dispatcher launched for the calling host thread's most recent md_conv2d (0 before any call, or when
 * the call returned without launching).  Diagnostic only: lets bench.py attribute per-launch HIP-event timings. */
enum {
    MD_CONV_KERNEL_NONE = 0,
    MD_CONV_KERNEL_PINGPONG = 1,        /* conv_pingpong_kernel: 256x256x64, 8 waves, MFMA-bound layers */
    MD_CONV_KERNEL_IGEMM_128 = 2,       /* conv_igemm_kernel 128x128, LDS-DMA, Cin % 64 == 0 */
    MD_CONV_KERNEL_IGEMM_SMALL_COUT = 3,/* conv_igemm_kernel 64- / 32-cout tiles */
    MD_CONV_KERNEL_IGEMM_GENERIC_K = 4, /* conv_igemm_kernel generic K walk (the 7x7 stem) */
    MD_CONV_KERNEL_HALO = 5,            /* conv3x3_halo_kernel (3x3 layers with Cout <= 64; variant 11 / 27) */
    MD_CONV_KERNEL_OTHER = 6,           /* A/B variants */
    MD_CONV_KERNEL_BOTTLENECK = 7,      /* bottleneck64_kernel (md_bottleneck) */
    MD_CONV_KERNEL_STREAM_1X1 = 8,      /* conv1x1_stream_kernel: weight-stationary pointwise layers, K <= 512 */
    MD_CONV_KERNEL_C3_PAIR = 9          /* c3pair_kernel (md_c3_pair) */
};
int md_conv2d_last_kernel(void);
/* Running count of the kernels the calling host thread's conv-family calls (md_conv2d, md_conv2d_head, md_conv1x1_dual,
 * md_bottleneck) have launched: a call on a batch past the chunk limit launches once per image chunk.  Diagnostic only:
 * bench.py divides a call's algorithmic work and bracketed time by its launches. */
long long md_conv2d_launch_count(void);

/* ------------------------------------------------------------------------------------------
 * Streaming NHWC bf16 helpers between convs
 * ------------------------------------------------------------------------------------------ */
typedef struct md_pool_attrs {
    int32_t k, stride, pad;
    int32_t zero_pad; /* 1: out-of-image taps count as 0 (explicit zero Pad + MaxPool2d,
                         centernet/src/resnet.py:199-204); 0: ignored (-inf padding) */
} md_pool_attrs;
/* in x[N,H,W,C] bf16 ; out y[N,Ho,Wo,C] bf16.  extra: md_pool_attrs (required). */
int md_maxpool2d(MD_AOT_ARGS);

/* The pooling chain of an SPPF block (build-authored YOLOv5 / YOLOv8 models; SURVEY 0.2: the reference names the families only) in ONE
 * launch, in place on the block's concat buffer: with x = buf[.., 0:C],
 *   buf[.., C:2C] = mp(x), buf[.., 2C:3C] = mp(mp(x)), buf[.., 3C:4C] = mp(mp(mp(x))),  mp = max-pool k x k, stride 1, pad k/2, out-of-image
 * taps ignored (torch semantics) -- bit-identical to three md_maxpool2d(zero_pad = 0) launches + the copies into the buffer.
 * in/out: buf[N,H,W,Ctot] bf16 (Ctot >= 4 C, both multiples of 8).  extra: md_sppf_attrs (required).
 * MD_ERR_SIZE when an image's working set does not fit LDS (md_sppf_pool_groups(H, W, C) == 0): run the three pools instead. */
typedef struct md_sppf_attrs {
    int32_t channels;   /* C */
    int32_t k;          /* window (odd) */
} md_sppf_attrs;
int md_sppf_pool(MD_AOT_ARGS);
int md_sppf_pool_groups(int H, int W, int C);
/* YOLOv8 Detect decode (anchor-free, distribution focal loss bins; Ultralytics v8 convention; absent from the reference,
 * BASELINE configs[3]: parity unpinned).  in head[B,H,W,Cp] bf16: channels [0, 4*reg_max) = (l,t,r,b) x reg_max bin logits,
 * [4*reg_max, 4*reg_max + nc) = class logits ; out boxes[B,total,4] f32 (xyxy, pixels), scores[B,total] f32 (max class
 * probability, -FLT_MAX at or below conf_thres), labels[B,total] i32; this level fills rows [out_offset, out_offset + H*W). */
typedef struct md_yolov8_attrs {
    int32_t num_classes, reg_max;
    float stride, conf_thres;
    int32_t out_offset, out_total;
} md_yolov8_attrs;
int md_yolov8_decode(MD_AOT_ARGS);

/* Mask R-CNN: per-detection class channel of the mask head + sigmoid (absent from the reference: standard head, parity
 * unpinned).  in logits[R,S,S,Cpad] bf16, dets[R,6] f32 (x1,y1,x2,y2,score,label) ; out masks[R,S,S] f32 (zeros for empty
 * detection slots).  extra: int32 num_classes. */
int md_mask_select(MD_AOT_ARGS);

/* Mask R-CNN: paste each detection's S x S mask into the image (the standard last step of the mask branch, BASELINE configs[4]):
 * bilinear resampling of the mask probabilities onto the pixel centres of the detection's box (grid_sample align_corners=False, zero
 * padding: mmdet _do_paste_mask / torchvision paste_masks_in_image), then mask >= threshold.  Absent from the reference (README bullet):
 * parity unpinned; oracle/np_ops.py::paste_masks restates the kernel's fp32 operation sequence, the comparison is bit-exact.
 * in masks[R,S,S] f32, dets[R,6] f32 (x1,y1,x2,y2,score,label; slots with score 0 or an empty box give an all-zero mask) ;
 * out bits != 0: words[R, img_h, ceil(img_w / 32)] i32 (bit j of word k of a row = pixel 32 k + j), else mask[R, img_h, img_w] u8 (0 / 1).
 * One HBM-write-bound launch: img_h * img_w / 8 bytes per detection (bit form). */
typedef struct md_paste_attrs {
    int32_t img_h, img_w;
    float threshold;   /* 0.5 in the public implementations */
    int32_t bits;      /* 1: bit mask (32 pixels per int32 word), 0: uint8 mask */
} md_paste_attrs;
int md_paste_masks(MD_AOT_ARGS);

/* Anchor target assignment on the device.  Replaces create_target_np
 * (minddet/models/pointpillars/src/core/target_assigner.py:29-166) as TargetAssigner.assign drives it (:196-224) with
 * positive_fraction None: similarity = iou_jit(rbbox2d_to_near_bbox(boxes[:, [0,1,3,4,6]]), eps 0)
 * (region_similarity.py:46-59), encoding = second_box_encode (box_np_ops.py:8-37).
 * in : anchors[A,7] f32, gt_boxes[G,7] f32, gt_classes[G] i32 (>= 1), matched_thr[A] f32, unmatched_thr[A] f32,
 *      anchors_mask[A] u8 or NULL (NULL = all inside)
 * out: labels[A] i32 (-1 ignore / outside the mask, 0 background, class), bbox_targets[A,7] f32,
 *      bbox_outside_weights[A] f32 (1 on foreground), gt_ids[A] i32 (matched ground truth, -1 elsewhere)
 * optional trailing workspace (4*G bytes).  G <= 1024.  labels / gt_ids bit-exact vs the reference; log() targets <= 4 ulp. */
int md_assign_targets(MD_AOT_ARGS);

/* Image pre-processing on the device: bilinear affine warp with constant-0 border + normalisation +
 * layout.  Replaces cv2.resize + cv2.warpAffine(INTER_LINEAR) + (img / 255 - mean) / std of
 * minddet/models/centernet/src/dataset.py:223-256 and the ImagePreProcess cell of centernet_det.py:240-262.
 * in : img[N,Hs,Ws,3] uint8 ; mat[N,6] f32 = per image the 2x3 matrix mapping OUTPUT pixel (x, y) to SOURCE pixel
 *      (sx = m0 x + m1 y + m2, sy = m3 x + m4 y + m5) ; norm[6] f32 = mean[3], std[3] of the 0..1 image
 * out: y[N, pad_lo + out_h + pad_hi, pad_lo + out_w + pad_hi, C] bf16, C = 4 (stem layout: pad_lo 7, pad_hi 9) or 8
 *      (pad 0); channels 3.. and the border are zero.
 * fp32 interpolation (cv2's 1/32-pixel fixed point is not reproduced; cv2 is absent here: parity unpinned). */
typedef struct md_preprocess_attrs {
    int32_t out_h, out_w, pad_lo, pad_hi;
} md_preprocess_attrs;
int md_image_preprocess(MD_AOT_ARGS);

/* Modulated deformable convolution (DCNv2), step 1: deformable im2col.  Replaces ops.deformable_conv2d as wrapped by
 * ModulatedDeformConv2d (minddet/models/centernet/src/resnet.py:24-106; CenterNet neck, centernet_det.py:123-160):
 * in  x[N,H,W,C] bf16 ; off[N,Ho,Wo,Coff >= 3*k*k] bf16 = the offset conv's output in the wrapper's own channel order
 *     (tap t = ky*k + kx: channel 2t = dy, 2t+1 = dx; channel 2*k*k + t = mask logit)
 * out cols[N,Ho,Wo,k*k*C] bf16, cols[.., t*C + c] = sigmoid(mask_t) * bilinear(x[.., c]; ho*s - p + ky + dy_t, wo*s - p + kx + dx_t),
 *     zero outside the image.  extra: md_pool_attrs (k, stride, pad).
 * Step 2 is md_conv2d with kh = kw = 1 on `cols` and the layer's [Cout][k*k*C] weights (K order (tap, channel) = the packed
 * 3x3 layout).  The arithmetic of the MindSpore primitive is not in the reference: published DCNv2 definition, parity unpinned. */
int md_deform_cols(MD_AOT_ARGS);

/* The ResNet stem in one launch: conv 7x7 / s2 / p3 (3 -> 64) + folded BN + ReLU + zero-pad + MaxPool 3x3 / s2
 * (minddet/models/centernet/src/resnet.py:199-204 and :226-233; the stem of every ResNet graph of the reference).
 * in : x[N, H+16, W+16, 4] bf16 -- the image in the STEM LAYOUT: channels (c0, c1, c2, 0), a zero border of
 *      md_stem_layout_pad(0) = 7 pixels on the left / top and md_stem_layout_pad(1) = 9 on the right / bottom;
 *      w[64, 224] bf16 -- BN-folded weights, K = (ky 0..6, kx 0..7, c 0..3) with kx 7 and c 3 zero; bias[64] f32
 * out: y[N, H/4, W/4, 64] bf16.   H % 16 == 0 and W % 64 == 0 (else MD_ERR_ARG: use md_conv2d + md_maxpool2d). */
int md_stem_pool(MD_AOT_ARGS);

/* The first convolution of the one-stage detectors on the same STEM LAYOUT input: 3 input channels, stride 2, folded BN, activation --
 * YOLOv5 v6 stem 6x6 / s2 / p2 and YOLOv8 stem 3x3 / s2 / p1 (configs/yolov5, configs/yolov8; conv + BN + activation cells like
 * minddet/models/centernet/src/resnet.py:199-204).  Weights in registers, the raw input patch in LDS (no im2col, padding taps read
 * real zeros), 4-channel input: see csrc/stemconv.hip.
 * in : x[N, H+16, W+16, 4] bf16 (stem layout), w[COUT, K] bf16 with K = (ky, kx', c 0..3), c 3 zero --
 *        kh = 6: K = 192, kx' = kx + 1 in 0..7 (columns 0 and 7 zero: the 6-tap window runs as an aligned 8-tap window);
 *        kh = 3: K = 48,  kx' = kx in 0..3 (column 3 zero);
 *      bias[COUT] f32 ; out: y[N, H/2, W/2, COUT] bf16, COUT = 32 or 64.  H % 16 == 0 and W % 64 == 0 (else MD_ERR_ARG: use md_conv2d). */
typedef struct md_stem_conv_attrs {
    int32_t kh;   /* 6 or 3 (square kernel, stride 2, pad kh / 2 - (kh == 6)) */
    int32_t act;  /* 0 none, 1 ReLU, 2 SiLU */
} md_stem_conv_attrs;
int md_stem_conv(MD_AOT_ARGS);
int md_stem_layout_pad(int which);

/* FPN top-down step: in lateral[N,H,W,C], top[N,Ht,Wt,C] bf16 ; out y = lateral + nearest_up(top). */
int md_upsample_add(MD_AOT_ARGS);
typedef struct md_slice_attrs {
    int32_t c0, width;
} md_slice_attrs;
/* in x[..., C] bf16 ; out y[..., width] f32 = x[..., c0:c0+width].  extra: md_slice_attrs. */
int md_slice_cast(MD_AOT_ARGS);

/* copy a tensor into a channel slice of a wider one (channel concat of tensors that were not written in place):
 * in src[N,H,W,C] bf16 ; out dst[N,H,W,Ctot] bf16 (only channels [c0, c0+C) are written).  extra: md_slice_attrs
 * (c0 = destination offset, width = C) */
int md_concat_copy(MD_AOT_ARGS);
/* nearest 2x upsample of a channel slice into a channel slice: in src[N,H,W,Cs] bf16 (channels [src_c0, src_c0 + width) are read) ;
 * out dst[N,2H,2W,Ctot] bf16 (channels [c0, c0 + width) are written).  extra: md_upsample2x_attrs (required) */
typedef struct md_upsample2x_attrs {
    int32_t c0, width;   /* destination offset, channels copied */
    int32_t src_c0;      /* first source channel */
} md_upsample2x_attrs;
int md_upsample2x(MD_AOT_ARGS);
/* in x[N,H,W,C] bf16 ; out y[N,width,H,W] f32 (NCHW, the layout centernet/src/decode.py consumes) */
int md_nhwc_to_nchw_f32(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * Anchor / prior generation
 * ------------------------------------------------------------------------------------------ */
typedef struct md_fpn_anchor_attrs {
    int32_t num_levels, num_ratios;
    int32_t feat_h[8], feat_w[8], stride[8];
    float scale;
    float ratios[16];
} md_fpn_anchor_attrs;
/* out anchors[sum_l H_l*W_l*A, 4] f32, levels concatenated, locations row-major, ratio fastest.
 * Absent from the reference (dagger) -- mmdet AnchorGenerator convention. */
int md_anchors_fpn(MD_AOT_ARGS);

typedef struct md_anchor3d_attrs {
    int32_t feat_h, feat_w, num_rot;
    double range[6];      /* anchor_range x0,y0,z0,x1,y1,z1 */
    double z_offset;      /* anchor_offsets[2] */
    double size[3];       /* one size triple (w,l,h) */
    double rotations[8];
    int32_t slot_off, slots_total; /* slots_total > 0: this generator fills rows [slot_off, slot_off + R) of the [.., slots_total, 7]
                                      anchor table of each location -- TargetAssigner.generate_anchors' concat of several generators
                                      on axis -2 (pointpillars/src/core/target_assigner.py:227-249) written in place; 0 = standalone */
} md_anchor3d_attrs;
/* create_anchors_3d_stride (pointpillars/src/core/box_np_ops.py:453-523) for ONE size:
 * out anchors[1,H,W,1,R,7] f32 (or [1,H,W,slots_total,7]), bit-identical to the numpy result (np.arange float32 fill). */
int md_anchors_3d_stride(MD_AOT_ARGS);

typedef struct md_anchor3d_range_attrs {
    int32_t feat_d, feat_h, feat_w, num_sizes /* <= 4 */, num_rot /* <= 8 */;
    int32_t linspace_mode;  /* np.linspace arithmetic: 0 = float32 products and sums (numpy >= 2, what the reference's code computes
                               on a current numpy: pinned by tests/golden/anchors_range_vectors.npz), 1 = float64 rounded once
                               (numpy 1.21, the reference's requirements pin; restated from numpy's source, parity unpinned) */
    int32_t slot_off, slots_total; /* as in md_anchor3d_attrs */
    double range[6];        /* anchor_range x0,y0,z0,x1,y1,z1 */
    double sizes[4][3];
    double rotations[8];
} md_anchor3d_range_attrs;
/* create_anchors_3d_range (pointpillars/src/core/box_np_ops.py:526-568): out anchors[D,H,W,S,R,7] f32 (or [D,H,W,slots_total,7]),
 * centres on np.linspace(range_lo, range_hi, n) per axis (the last one exactly range_hi). */
int md_anchors_3d_range(MD_AOT_ARGS);

typedef struct md_anchor_mask_attrs {
    int32_t grid_x, grid_y;
    float voxel_x, voxel_y, offset_x, offset_y;
    float area_threshold;
} md_anchor_mask_attrs;
/* pointpillars/src/data/preprocess.py:211-225 + box_np_ops.py:745-776:
 * in coors[V,3] i32 (z,y,x), anchors_bv[N,4] f32 ; out area[N] f32, mask[N] u8 (area > threshold);
 * optional workspace grid_x*grid_y*4 bytes. */
int md_anchor_mask(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * Box codecs
 * ------------------------------------------------------------------------------------------ */
/* second_box_decode (pointpillars/src/core/box_ops.py:47-85): in enc[...,7] f32, anchors[A,7] f32
 * (row i uses anchor i % A) ; out boxes[...,7] f32. */
int md_second_box_decode(MD_AOT_ARGS);
typedef struct md_delta2bbox_attrs {
    float means[4], stds[4];
    float max_ratio;      /* |log(wh_ratio_clip)| */
    float clip_w, clip_h; /* <= 0: no clipping */
} md_delta2bbox_attrs;
/* in rois[n,4] f32, deltas[n,4] f32 ; out boxes[n,4] f32.  Absent from the reference (dagger). */
int md_delta2bbox(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * Segmented top-k (ops.TopK(sorted=True) stated as a stable descending sort)
 * ------------------------------------------------------------------------------------------ */
typedef struct md_topk_attrs {
    int32_t k;        /* <= 4096 */
    float min_score;  /* only scores > min_score are selectable; -FLT_MAX to disable */
    int32_t max_segment; /* upper bound of the segment lengths if the caller knows it (seg_off lives on the
                            device); > 32768 switches to the multi-workgroup select. 0 = unknown */
} md_topk_attrs;
/* in scores[T] f32, seg_off[L+1] i32 ; out values[L,k] f32 (padded -FLT_MAX), indices[L,k] i32
 * (relative to the segment, padded 0), count[L] i32 ; optional workspace
 * (L*(2048*4+4) rounded up to 256, + L*8192*8 bytes) for the multi-workgroup path. */
int md_topk_segmented(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * RoIAlign over an FPN pyramid (absent from the reference, SURVEY a12: torchvision semantics)
 * ------------------------------------------------------------------------------------------ */
typedef struct md_roi_align_attrs {
    int32_t num_levels, pooled, sampling_ratio, aligned;
    int32_t k_min, canonical_level;
    float canonical_scale;
    float spatial_scale[6];
} md_roi_align_attrs;
/* in rois[R,5] f32 (batch_idx,x1,y1,x2,y2), feat_0..feat_{L-1} [N,H_l,W_l,C] bf16 ;
 * out pooled[R,P,P,C] bf16, level[R] i32 (pointer may be NULL). */
int md_roi_align(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * CenterNet head decode pieces (centernet/src/decode.py, utils.py:132-157)
 * ------------------------------------------------------------------------------------------ */
typedef struct md_clip_attrs {
    float lo, hi;
} md_clip_attrs;
/* sigmoid then clip to [lo,hi] (default 1e-4, 1-1e-4): in x f32 ; out y f32 */
int md_sigmoid_clip(MD_AOT_ARGS);
/* heat * (heat == maxpool3x3_same(heat)): in heat[B,C,H,W] f32 ; out same shape */
int md_heat_nms(MD_AOT_ARGS);
typedef struct md_heat_peaks_attrs {
    int32_t c0, num_classes; /* heat-map channels [c0, c0 + num_classes) of the head (c0 % 8 == 0) */
    float lo, hi;            /* clip range of md_sigmoid_clip */
} md_heat_peaks_attrs;
/* The CenterNet head's heat-map post-processing in one pass, same arithmetic per element as the three ops it replaces
 * (centernet/src/centernet_det.py:374-399 sigmoid + clip of the `hm` head, centernet/src/decode.py:40-64 `NMS` = 3x3 max-pool peak
 * test): in head[B,H,W,Cp] bf16 NHWC ; out heat[B,num_classes,H,W] f32 NCHW (sigmoid-clipped value where it is the maximum of its
 * 3x3 neighbourhood, else 0), hm[B,num_classes,H,W] f32 (sigmoid-clipped values; pointer may be NULL). extra: md_heat_peaks_attrs. */
int md_heat_peaks(MD_AOT_ARGS);
/* in top_score[B,K] f32, top_ind2[B,K] i32 (index into C*K), cls_inds[B,C,K] i32, wh[B,2,H,W] f32,
 *    reg[B,2,H,W] f32 or NULL ; out det[B,K,6] f32 (x1,y1,x2,y2,score,cls), inds[B,K] i32, cls[B,K] i32 */
int md_centernet_assemble(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * Two-stage (Faster R-CNN style) glue between conv stacks and detection ops.  Absent from the
 * reference; mmdet/torchvision public conventions; pinned by oracle/pipeline.py only.
 * ------------------------------------------------------------------------------------------ */
typedef struct md_rpn_decode_attrs {
    int32_t num_anchors;          /* A: head channels [0,A) objectness, [A,5A) deltas (a*4+j) */
    md_delta2bbox_attrs decode;
} md_rpn_decode_attrs;
/* in head[B,H,W,Cp] bf16, anchors[H*W*A,4] f32, idx[B,k] i32, cnt[B] i32 ;
 * out boxes[B,k,4] f32 (0 past cnt), scores[B,k] f32 = sigmoid(logit) (-FLT_MAX past cnt) */
int md_rpn_decode(MD_AOT_ARGS);
/* in boxes[L,B,k,4] f32, scores[L,B,k] f32, keep[L,B,k] u8 ; out mboxes[B,L*k,4], mscores[B,L*k]
 * (score -FLT_MAX where suppressed) */
int md_rpn_merge(MD_AOT_ARGS);
/* in mboxes[B,P,4] f32, topv[B,post] f32, topi[B,post] i32, cnt[B] i32 ;
 * out rois[B*post,5] f32 (batch_idx,x1,y1,x2,y2; zero box past cnt), roi_scores[B*post] f32 */
int md_make_rois(MD_AOT_ARGS);
typedef struct md_rcnn_attrs {
    int32_t num_classes;  /* nc foreground classes; logits [0,nc] with background LAST */
    int32_t reg_offset;   /* first channel of the 4*nc class-specific deltas */
    float score_thr;
    md_delta2bbox_attrs decode;
} md_rcnn_attrs;
/* in cls_reg[R,Cp] bf16, roi_cnt[B] i32 ; out cand[B, (R/B)*nc] f32 = softmax prob if > score_thr
 * and the RoI slot is valid, else -FLT_MAX */
int md_rcnn_scores(MD_AOT_ARGS);
/* in cls_reg[R,Cp] bf16, rois[R,5] f32, sel_idx[B,npre] i32 (j*nc + c), sel_cnt[B] i32 ;
 * out boxes[B,npre,4] f32, labels[B,npre] i32 (-1 past cnt) */
int md_rcnn_decode_selected(MD_AOT_ARGS);
/* class-wise merge + packing (shape of centernet/src/post_process.py:36-61):
 * in boxes[B,npre,4] f32, scores[B,npre] f32, labels[B,npre] i32, keep_idx[B,npre] i32, num[B] i32 ;
 * out dets[B,max_det,6] f32 (x1,y1,x2,y2,score,label; zero padded), count[B] i32.
 * 9-parameter form: in ... num, sel_cnt[B] i32 (candidates the pre-NMS top-k selected) ; out dets, count, status[B] i32
 * (IN/OUT, caller-cleared): bit 0 is OR-ed in when the NMS saw a FULL prefix (sel_cnt >= npre) and found fewer than
 * max_det survivors -- the one case in which the top-npre cut can change the result of the untruncated class-wise NMS
 * (DESIGN.md "pre-NMS prefix"); clear = provably identical to running the NMS on every candidate. */
int md_pack_detections(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * CenterPoint head decode (centerpoint/det3d_ms/models/bbox_heads/center_head.py:297-345,398-430)
 * ------------------------------------------------------------------------------------------ */
typedef struct md_centerpoint_attrs {
    int32_t off_reg, off_height, off_dim, off_rot, off_vel /* -1: none */, off_hm, num_classes;
    float score_threshold, out_size_factor;
    float voxel_size[2], pc_range[2], post_center_range[6];
} md_centerpoint_attrs;
/* in head[B,H,W,C] bf16 (one task's attributes at the given channel offsets) ;
 * out scores[B,HW] f32 (-1 where masked), labels[B,HW] i32 (-1), boxes[B,HW,9] f32 (x,y,z,dx,dy,dz,vx,vy,rot; 0 where
 * masked), nms_boxes[B,HW,7] f32 (x,y,z,dy,dx,dz,-rot-pi/2 -- the operand of boxes_iou_nms_gpu, center_head.py:426-430) */
int md_centerpoint_decode(MD_AOT_ARGS);
/* in src[B,n,W] f32, idx[B,k] i32, cnt[B] i32 or NULL ; out out[B,k,W] f32 (zero rows past cnt) */
int md_gather_rows(MD_AOT_ARGS);

/* rotated BEV boxes -> standup (axis-aligned) boxes: pointpillars/src/core/box_np_ops.py:316-341,172-177
 * (call site pointpillars/src/predict.py:61-78).  in boxes[N,5] (x,y,dx,dy,r) or [N,7] (x,y,z,dx,dy,dz,r) f32 ;
 * out standup[N,4] f32 (xmin,ymin,xmax,ymax) */
int md_standup_boxes(MD_AOT_ARGS);

/* ------------------------------------------------------------------------------------------
 * YOLOv5 Detect decode (absent from the reference; Ultralytics v6/v7 convention; parity unpinned)
 * ------------------------------------------------------------------------------------------ */
typedef struct md_yolo_attrs {
    int32_t num_classes, num_anchors;   /* head channels per anchor = 5 + num_classes, anchor-major */
    float stride;
    float anchors[6];                   /* (w,h) in pixels for up to 3 anchors of this level */
    float conf_thres;
    int32_t out_offset;                 /* first row of this level inside the per-image outputs */
    int32_t out_total;                  /* rows per image over all levels */
} md_yolo_attrs;
/* in head[B,H,W,Cp] bf16 ; out boxes[B,out_total,4] f32 (x1,y1,x2,y2), scores[B,out_total] f32 = obj*max_cls
 * (-FLT_MAX if obj <= conf_thres or score <= conf_thres), labels[B,out_total] i32 ; only rows
 * [out_offset, out_offset + H*W*A) are written */
int md_yolo_decode(MD_AOT_ARGS);

#ifdef __cplusplus
}
#endif
#endif /* MINDDET_HIP_H_ */
