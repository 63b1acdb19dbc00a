"""Host-side mirror of the reference's detection-op interfaces, bound to libminddet_hip.so.

Names, argument meaning and return conventions follow the reference so that its call sites
read the same (paths relative to /root/reference/minddet/models):

  NMS                     centerpoint/det3d_ms/ops/nms_cpu.py:7-27     (boxes[N,7], thresh) -> (keep[N] i32, num)
  BoxesIouBevGpu, BoxesOverlapBevGpu, NumGpu, NmsNormalGpu
                          centerpoint/det3d_ms/ops/test_custom_pytorch/iou_gpu.py:14-81
  boxes_iou_bev, boxes_iou3d_gpu, nms_gpu, nms_normal_gpu
                          centerpoint/det3d_ms/ops/iou3d_nms/iou3d_nms_utils.py:12-116
  iou_jit                 pointpillars/src/core/box_np_ops.py:639-679
  nms_jit / apply_nms     pointpillars/src/core/nms.py:7-41,85-112     (via nms_aligned)
  circle_nms              centerpoint/det3d_ms/core/utils/circle_nms_jit.py:6-36

All tensors are torch CUDA tensors; work is enqueued on the current stream; nothing here
synchronises.  There is no CPU path.
"""
import ctypes
import math

import torch

from . import _lib


class _IouAttrs(ctypes.Structure):
    _fields_ = [("eps", ctypes.c_float)]


class _NmsAttrs(ctypes.Structure):
    _fields_ = [("iou_threshold", ctypes.c_float), ("eps", ctypes.c_float), ("mode", ctypes.c_int32),
                ("max_output", ctypes.c_int32)]


NMS_MODE_JIT = 0      # nms_jit: suppress iff ovr >= thr, eps on w/h/area
NMS_MODE_PLUS1 = 1    # apply_nms: +1 pixel convention, suppress iff ovr > thr
NMS_MODE_STRICT = 2   # suppress iff IoU > thr, union clamped at 1e-8


def _f32c(t):
    return t.contiguous().to(torch.float32)


def _thresh_tensor(thresh, device):
    if isinstance(thresh, torch.Tensor):
        return thresh.to(device=device, dtype=torch.float32).reshape(1).contiguous()
    return torch.full((1,), float(thresh), dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------- AOT-op "cells"
class NMS:
    """Mirror of det3d_ms.ops.nms_cpu.NMS, executed on the GPU (boxes_iou_nms_gpu)."""

    def __call__(self, boxes, thresh):
        boxes = _f32c(boxes)
        n = boxes.shape[0]
        keep = torch.empty((n,), dtype=torch.int32, device=boxes.device)
        num = torch.empty((1,), dtype=torch.int32, device=boxes.device)
        _lib.call("boxes_iou_nms_gpu", [boxes, _thresh_tensor(thresh, boxes.device), keep, num])
        return keep, num[0]

    construct = __call__


class _PairMatrix:
    _sym = None

    def __call__(self, a, b):
        a, b = _f32c(a), _f32c(b)
        out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
        _lib.call(self._sym, [a, b, out])
        return out

    construct = __call__


class BoxesIouBevGpu(_PairMatrix):
    _sym = "BoxesIouBevGpu"


class BoxesOverlapBevGpu(_PairMatrix):
    _sym = "BoxesOverlapBevGpu"


class _NmsCell:
    _sym = None

    def __call__(self, boxes, thresh):
        boxes = _f32c(boxes)
        n = boxes.shape[0]
        keep = torch.empty((n,), dtype=torch.int64, device=boxes.device)
        num = torch.empty((1,), dtype=torch.int32, device=boxes.device)
        _lib.call(self._sym, [boxes, _thresh_tensor(thresh, boxes.device), keep, num])
        return keep, num

    construct = __call__


class NumGpu(_NmsCell):  # (sic) the reference's class name for NmsGpu, iou_gpu.py:51
    _sym = "NmsGpu"


NmsGpu = NumGpu


class NmsNormalGpu(_NmsCell):
    _sym = "NmsNormalGpu"


# ----------------------------------------------------------------------------- iou3d_nms_utils mirror
def boxes_iou_bev(boxes_a, boxes_b):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return BoxesIouBevGpu()(boxes_a, boxes_b)


def boxes_overlap_bev(boxes_a, boxes_b):
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    return BoxesOverlapBevGpu()(boxes_a, boxes_b)


def to_pcdet(boxes):
    boxes = boxes[:, [0, 1, 2, 4, 3, 5, -1]].clone()
    boxes[:, -1] = -boxes[:, -1] - math.pi / 2
    return boxes


def boxes_iou3d_gpu(boxes_a, boxes_b):
    """iou3d_nms_utils.py:40-81: BEV overlap x height overlap / volume union."""
    assert boxes_a.shape[1] == boxes_b.shape[1] == 7
    boxes_a, boxes_b = to_pcdet(boxes_a), to_pcdet(boxes_b)
    a_max = (boxes_a[:, 2] + boxes_a[:, 5] / 2).view(-1, 1)
    a_min = (boxes_a[:, 2] - boxes_a[:, 5] / 2).view(-1, 1)
    b_max = (boxes_b[:, 2] + boxes_b[:, 5] / 2).view(1, -1)
    b_min = (boxes_b[:, 2] - boxes_b[:, 5] / 2).view(1, -1)
    overlaps_bev = boxes_overlap_bev(boxes_a.contiguous(), boxes_b.contiguous())
    overlaps_h = torch.clamp(torch.min(a_max, b_max) - torch.max(a_min, b_min), min=0)
    overlaps_3d = overlaps_bev * overlaps_h
    vol_a = (boxes_a[:, 3] * boxes_a[:, 4] * boxes_a[:, 5]).view(-1, 1)
    vol_b = (boxes_b[:, 3] * boxes_b[:, 4] * boxes_b[:, 5]).view(1, -1)
    return overlaps_3d / torch.clamp(vol_a + vol_b - overlaps_3d, min=1e-6)


def _sorted_order(scores):
    # stable descending sort: ties keep the lower index first (SURVEY 8c TopK definition)
    return torch.sort(scores, dim=0, descending=True, stable=True)[1]


def nms_gpu(boxes, scores, thresh, pre_maxsize=None, **kwargs):
    """iou3d_nms_utils.py:84-99: returns (indices into the ORIGINAL boxes, None)."""
    assert boxes.shape[1] == 7
    order = _sorted_order(scores)
    if pre_maxsize is not None:
        order = order[:pre_maxsize]
    keep, num = NumGpu()(boxes[order].contiguous(), thresh)
    return order[keep[: int(num.item())]].contiguous(), None


def nms_normal_gpu(boxes, scores, thresh, **kwargs):
    """iou3d_nms_utils.py:102-116."""
    assert boxes.shape[1] == 7
    order = _sorted_order(scores)
    keep, num = NmsNormalGpu()(boxes[order].contiguous(), thresh)
    return order[keep[: int(num.item())]].contiguous(), None


# ----------------------------------------------------------------------------- axis-aligned
def iou_jit(boxes, query_boxes, eps=0.0):
    boxes, query_boxes = _f32c(boxes), _f32c(query_boxes)
    out = torch.empty((boxes.shape[0], query_boxes.shape[0]), dtype=torch.float32, device=boxes.device)
    _lib.call("md_iou_aligned", [boxes, query_boxes, out], extra=_IouAttrs(float(eps)))
    return out


def nms_aligned(boxes_sorted, thresh, eps=0.0, mode=NMS_MODE_JIT, count=None, group=None, max_output=0,
                workspace=None):
    """Greedy NMS over score-sorted corner boxes [N,4] or a batch [B,N,4].

    Returns (keep_mask u8, keep_idx i32 zero-padded, num i32[B])."""
    b = _f32c(boxes_sorted)
    batched = b.dim() == 3
    B, n = (b.shape[0], b.shape[1]) if batched else (1, b.shape[0])
    dev = b.device
    mask = torch.empty((B, n) if batched else (n,), dtype=torch.uint8, device=dev)
    idx = torch.empty((B, n) if batched else (n,), dtype=torch.int32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    if count is not None:
        count = count.to(device=dev, dtype=torch.int32).reshape(B).contiguous()
    if group is not None:
        group = group.to(device=dev, dtype=torch.int32).contiguous()
    params = [b, count, group, mask, idx, num]
    if workspace is None and B * n > 0:
        # suppression-mask scratch from torch's caching allocator (never blocks), not from the op's hipMallocAsync fallback
        cb = (n + 63) // 64
        workspace = torch.empty((B * n * cb * 8 + (B * 4 + 255) // 256 * 256,), dtype=torch.uint8, device=dev)
    if workspace is not None:
        params.append(workspace)
    _lib.call("md_nms_aligned", params, extra=_NmsAttrs(float(thresh), float(eps), int(mode), int(max_output)))
    return mask, idx, num


def nms_jit(dets, thresh, eps=0.0):
    """pointpillars/src/core/nms.py:85-112 on device: dets [N,5] (x1,y1,x2,y2,score) -> keep
    indices (original numbering, score order), as a device int64 tensor."""
    order = _sorted_order(dets[:, 4])
    _, idx, num = nms_aligned(dets[order, :4].contiguous(), thresh, eps, NMS_MODE_JIT)
    return order[idx[: int(num.item())].long()]


def circle_nms(dets, thresh):
    """dets [N,3] = x, y, score. Returns keep indices in the original numbering."""
    order = _sorted_order(dets[:, 2])
    xy = _f32c(dets[order, :2])
    n = xy.shape[0]
    mask = torch.empty((n,), dtype=torch.uint8, device=xy.device)
    idx = torch.empty((n,), dtype=torch.int32, device=xy.device)
    num = torch.empty((1,), dtype=torch.int32, device=xy.device)
    _lib.call("md_circle_nms", [xy, _thresh_tensor(thresh, xy.device), mask, idx, num])
    return order[idx[: int(num.item())].long()]


# ----------------------------------------------------------------------------- anchors / codecs / select / roialign
class _FpnAnchorAttrs(ctypes.Structure):
    _fields_ = [("num_levels", ctypes.c_int32), ("num_ratios", ctypes.c_int32), ("feat_h", ctypes.c_int32 * 8),
                ("feat_w", ctypes.c_int32 * 8), ("stride", ctypes.c_int32 * 8), ("scale", ctypes.c_float),
                ("ratios", ctypes.c_float * 16)]


class _Anchor3dAttrs(ctypes.Structure):
    _fields_ = [("feat_h", ctypes.c_int32), ("feat_w", ctypes.c_int32), ("num_rot", ctypes.c_int32),
                ("range", ctypes.c_double * 6), ("z_offset", ctypes.c_double), ("size", ctypes.c_double * 3),
                ("rotations", ctypes.c_double * 8), ("slot_off", ctypes.c_int32), ("slots_total", ctypes.c_int32)]


class _Anchor3dRangeAttrs(ctypes.Structure):
    _fields_ = [("feat_d", ctypes.c_int32), ("feat_h", ctypes.c_int32), ("feat_w", ctypes.c_int32), ("num_sizes", ctypes.c_int32),
                ("num_rot", ctypes.c_int32), ("linspace_mode", ctypes.c_int32), ("slot_off", ctypes.c_int32),
                ("slots_total", ctypes.c_int32), ("range", ctypes.c_double * 6), ("sizes", (ctypes.c_double * 3) * 4),
                ("rotations", ctypes.c_double * 8)]


class _AnchorMaskAttrs(ctypes.Structure):
    _fields_ = [("grid_x", ctypes.c_int32), ("grid_y", ctypes.c_int32), ("voxel_x", ctypes.c_float),
                ("voxel_y", ctypes.c_float), ("offset_x", ctypes.c_float), ("offset_y", ctypes.c_float),
                ("area_threshold", ctypes.c_float)]


class _DeltaAttrs(ctypes.Structure):
    _fields_ = [("means", ctypes.c_float * 4), ("stds", ctypes.c_float * 4), ("max_ratio", ctypes.c_float),
                ("clip_w", ctypes.c_float), ("clip_h", ctypes.c_float)]


class _TopkAttrs(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int32), ("min_score", ctypes.c_float), ("max_segment", ctypes.c_int32)]


class _RoiAlignAttrs(ctypes.Structure):
    _fields_ = [("num_levels", ctypes.c_int32), ("pooled", ctypes.c_int32), ("sampling_ratio", ctypes.c_int32),
                ("aligned", ctypes.c_int32), ("k_min", ctypes.c_int32), ("canonical_level", ctypes.c_int32),
                ("canonical_scale", ctypes.c_float), ("spatial_scale", ctypes.c_float * 6)]


class _ClipAttrs(ctypes.Structure):
    _fields_ = [("lo", ctypes.c_float), ("hi", ctypes.c_float)]


FLT_MAX = 3.4028234663852886e38


def fpn_anchors(feat_sizes, strides=(4, 8, 16, 32, 64), scale=8.0, ratios=(0.5, 1.0, 2.0), device="cuda"):
    at = _FpnAnchorAttrs()
    at.num_levels, at.num_ratios, at.scale = len(feat_sizes), len(ratios), float(scale)
    total = 0
    for i, ((h, w), s) in enumerate(zip(feat_sizes, strides)):
        at.feat_h[i], at.feat_w[i], at.stride[i] = int(h), int(w), int(s)
        total += h * w * len(ratios)
    for i, r in enumerate(ratios):
        at.ratios[i] = float(r)
    out = torch.empty((total, 4), dtype=torch.float32, device=device)
    _lib.call("md_anchors_fpn", [out], extra=at)
    return out


def create_anchors_3d_stride(feature_size, sizes=(1.6, 3.9, 1.56), anchor_strides=(0.4, 0.4, 0.0),
                             anchor_offsets=(0.2, -39.8, -1.78), rotations=(0, math.pi / 2),
                             anchor_range=(0.0, -39.68, -3.0, 69.12, 39.68, 1.0), dtype=torch.float32, device="cuda", out=None, slot_off=0):
    """pointpillars/src/core/box_np_ops.py:453-523 on device: returns [1,H,W,1,R,7] fp32.
    (anchor_strides / x,y offsets are unused by the reference too: it derives the stride from
    anchor_range, :476-477.)  out [1,H,W,slots,7] + slot_off: write rows [slot_off, slot_off + R) of a concatenated table."""
    d, h, w = feature_size
    assert d == 1 and len(rotations) == 2, "the reference hard-codes one z slice and two rotations (:492)"
    at = _Anchor3dAttrs()
    at.feat_h, at.feat_w, at.num_rot = int(h), int(w), len(rotations)
    for i in range(6):
        at.range[i] = float(anchor_range[i])
    at.z_offset = float(anchor_offsets[2])
    flat = [float(v) for v in (sizes if not isinstance(sizes[0], (list, tuple)) else sizes[0])]
    for i in range(3):
        at.size[i] = flat[i]
    for i, r in enumerate(rotations):
        at.rotations[i] = float(r)
    if out is None:
        out = torch.empty((1, h, w, 1, len(rotations), 7), dtype=torch.float32, device=device)
    else:
        at.slot_off, at.slots_total = int(slot_off), int(out.shape[-2])
    _lib.call("md_anchors_3d_stride", [out], extra=at)
    return out


def create_anchors_3d_range(feature_size, anchor_range, sizes=(1.6, 3.9, 1.56), rotations=(0, math.pi / 2), device="cuda",
                            linspace_mode=0, out=None, slot_off=0):
    """pointpillars/src/core/box_np_ops.py:526-568 on device: [D,H,W,S,R,7] fp32, centres on np.linspace(lo, hi, n) per axis.
    linspace_mode: md_anchor3d_range_attrs (0 = the float32 arithmetic of numpy >= 2, 1 = numpy 1.21's float64)."""
    d, h, w = [int(v) for v in feature_size]
    flat = [float(v) for row in (sizes if isinstance(sizes[0], (list, tuple)) else [sizes]) for v in row]
    ns = len(flat) // 3
    at = _Anchor3dRangeAttrs()
    at.feat_d, at.feat_h, at.feat_w, at.num_sizes, at.num_rot, at.linspace_mode = d, h, w, ns, len(rotations), int(linspace_mode)
    for i in range(6):
        at.range[i] = float(anchor_range[i])
    for k in range(ns):
        for i in range(3):
            at.sizes[k][i] = flat[3 * k + i]
    for i, r in enumerate(rotations):
        at.rotations[i] = float(r)
    if out is None:
        out = torch.empty((d, h, w, ns, len(rotations), 7), dtype=torch.float32, device=device)
    else:
        at.slot_off, at.slots_total = int(slot_off), int(out.shape[-2])
    _lib.call("md_anchors_3d_range", [out], extra=at)
    return out


class AnchorGeneratorStride:
    """pointpillars/src/core/anchor_generator.py:6-63 (same constructor arguments and properties); generate() runs on the device."""

    def __init__(self, sizes=(1.6, 3.9, 1.56), anchor_strides=(0.4, 0.4, 1.0), anchor_offsets=(0.2, -39.8, -1.78),
                 rotations=(0, math.pi / 2), class_id=None, match_threshold=-1, unmatch_threshold=-1,
                 anchor_range=(0.0, -39.68, -3.0, 69.12, 39.68, 1.0)):
        self._sizes, self._anchor_strides, self._anchor_offsets, self._rotations = sizes, anchor_strides, anchor_offsets, rotations
        self.class_id, self.match_threshold, self.unmatch_threshold = class_id, match_threshold, unmatch_threshold
        self.anchor_range = anchor_range

    @property
    def num_anchors_per_localization(self):
        flat = self._sizes if isinstance(self._sizes[0], (list, tuple)) else [self._sizes]
        return len(self._rotations) * len(flat)

    def generate(self, feature_map_size, device="cuda", out=None, slot_off=0):
        return create_anchors_3d_stride(feature_map_size, self._sizes, self._anchor_strides, self._anchor_offsets, self._rotations,
                                        self.anchor_range, device=device, out=out, slot_off=slot_off)


def generate_anchors(anchor_generators, feature_map_size, device="cuda"):
    """TargetAssigner.generate_anchors (pointpillars/src/core/target_assigner.py:227-249): every generator's anchors reshaped to
    [*shape[:3], -1, 7] and concatenated on axis -2, with the per-anchor matched / unmatched thresholds of its generator.  Each
    generator writes its rows of the concatenated table in place (no concat copy).  The threshold vectors are ordered as the
    reference builds them: generator after generator (np.concatenate of np.full blocks), NOT in the anchor table's order."""
    d, h, w = [int(v) for v in feature_map_size]
    per = [g.num_anchors_per_localization for g in anchor_generators]
    out = torch.empty((d, h, w, sum(per), 7), dtype=torch.float32, device=device)
    off = 0
    for g, n in zip(anchor_generators, per):
        g.generate(feature_map_size, device=device, out=out, slot_off=off)
        off += n
    locs = d * h * w
    matched = torch.cat([torch.full((locs * n,), float(g.match_threshold), dtype=torch.float32) for g, n in zip(anchor_generators, per)])
    unmatched = torch.cat([torch.full((locs * n,), float(g.unmatch_threshold), dtype=torch.float32) for g, n in zip(anchor_generators, per)])
    return {"anchors": out, "matched_thresholds": matched.to(device), "unmatched_thresholds": unmatched.to(device)}


def anchors_mask(coors, grid_size_xy, anchors_bv, voxel_size, pc_range, area_threshold):
    """preprocess.py:211-225: (anchors_area f32, anchors_mask bool) for one sample."""
    at = _AnchorMaskAttrs(int(grid_size_xy[0]), int(grid_size_xy[1]), float(voxel_size[0]), float(voxel_size[1]),
                          float(pc_range[0]), float(pc_range[1]), float(area_threshold))
    coors = coors.to(torch.int32).contiguous()
    bv = _f32c(anchors_bv)
    area = torch.empty((bv.shape[0],), dtype=torch.float32, device=bv.device)
    mask = torch.empty((bv.shape[0],), dtype=torch.uint8, device=bv.device)
    _lib.call("md_anchor_mask", [coors, bv, area, mask], extra=at)
    return area, mask.bool()


def second_box_decode(box_encodings, anchors):
    enc, anc = _f32c(box_encodings), _f32c(anchors).reshape(-1, 7)
    out = torch.empty_like(enc)
    _lib.call("md_second_box_decode", [enc, anc, out])
    return out


def delta2bbox(rois, deltas, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), max_shape=None, wh_ratio_clip=16 / 1000):
    at = _DeltaAttrs()
    for i in range(4):
        at.means[i], at.stds[i] = float(means[i]), float(stds[i])
    at.max_ratio = abs(math.log(wh_ratio_clip))
    at.clip_h, at.clip_w = (float(max_shape[0]), float(max_shape[1])) if max_shape is not None else (0.0, 0.0)
    rois, deltas = _f32c(rois), _f32c(deltas)
    out = torch.empty_like(rois)
    _lib.call("md_delta2bbox", [rois, deltas, out], extra=at)
    return out


def topk_segmented(scores, seg_offsets, k, min_score=None, out_cnt=None, max_segment=None):
    """scores [T] f32, seg_offsets [L+1] i32 (device) -> (values [L,k], indices [L,k] i32, count [L])."""
    scores = _f32c(scores).reshape(-1)
    L = seg_offsets.numel() - 1
    vals = torch.empty((L, k), dtype=torch.float32, device=scores.device)
    idx = torch.empty((L, k), dtype=torch.int32, device=scores.device)
    cnt = out_cnt if out_cnt is not None else torch.empty((L,), dtype=torch.int32, device=scores.device)
    max_segment = int(scores.numel() if max_segment is None else max_segment)
    params = [scores, seg_offsets, vals, idx, cnt]
    if max_segment > 4 * 8192 and k <= 4096 and L <= 65535:
        # the multi-workgroup select takes scratch: hand it a block of torch's caching allocator.  Without it the op falls back to
        # hipMallocAsync / hipFreeAsync, which BLOCKED the host ~7 ms per call on ROCm 7.2 (r03 tools/host_enqueue.py: 49.6 of the
        # 51.9 ms the host spent enqueuing one Faster R-CNN step sat in these seven calls, the device queue running dry behind each)
        hist = ((L * (2048 * 4 + 4)) + 255) // 256 * 256
        params.append(torch.empty((hist + L * 8192 * 8,), dtype=torch.uint8, device=scores.device))
    _lib.call("md_topk_segmented", params,
              extra=_TopkAttrs(int(k), -FLT_MAX if min_score is None else float(min_score), max_segment))
    return vals, idx, cnt


def top_k(scores2d, k):
    """ops.TopK(sorted=True) on the last axis of a [L, n] tensor."""
    L, n = scores2d.shape
    off = torch.arange(0, (L + 1) * n, n, dtype=torch.int32, device=scores2d.device)
    v, i, _ = topk_segmented(scores2d.contiguous(), off, k, max_segment=n)
    return v, i


def roi_align(feats, rois, out_size=7, spatial_scales=None, sampling_ratio=2, aligned=True, k_min=2,
              canonical_level=4, canonical_scale=224.0, return_levels=False):
    """feats: list of [N,H,W,C] bf16 NHWC levels; rois [R,5] (batch_idx,x1,y1,x2,y2) -> [R,P,P,C] bf16."""
    L = len(feats)
    at = _RoiAlignAttrs()
    at.num_levels, at.pooled, at.sampling_ratio, at.aligned = L, int(out_size), int(sampling_ratio), int(aligned)
    at.k_min, at.canonical_level, at.canonical_scale = int(k_min), int(canonical_level), float(canonical_scale)
    for i in range(L):
        at.spatial_scale[i] = float(spatial_scales[i])
    rois = _f32c(rois)
    R, C = rois.shape[0], feats[0].shape[3]
    out = torch.empty((R, out_size, out_size, C), dtype=torch.bfloat16, device=rois.device)
    lv = torch.empty((R,), dtype=torch.int32, device=rois.device) if return_levels else None
    _lib.call("md_roi_align", [rois] + list(feats) + [out, lv], extra=at)
    return (out, lv) if return_levels else out


def sigmoid_clip(x, lo=1e-4, hi=1 - 1e-4):
    x = _f32c(x)
    y = torch.empty_like(x)
    _lib.call("md_sigmoid_clip", [x, y], extra=_ClipAttrs(lo, hi))
    return y


class _PeakAttrs(ctypes.Structure):
    _fields_ = [("c0", ctypes.c_int32), ("num_classes", ctypes.c_int32), ("lo", ctypes.c_float), ("hi", ctypes.c_float)]


def heat_peaks(head, c0, num_classes, with_hm=False, lo=1e-4, hi=1 - 1e-4):
    """head [B,H,W,Cp] bf16 NHWC -> (heat [B,nc,H,W] f32 = sigmoid-clipped heat map zeroed off its 3x3 maxima, hm or None):
    nhwc_to_nchw_f32 + sigmoid_clip + md_heat_nms in one md_heat_peaks launch, bit-identical to the three."""
    b, h, w, _ = head.shape
    heat = torch.empty((b, num_classes, h, w), dtype=torch.float32, device=head.device)
    hm = torch.empty_like(heat) if with_hm else None
    _lib.call("md_heat_peaks", [head, heat, hm], extra=_PeakAttrs(int(c0), int(num_classes), float(lo), float(hi)))
    return heat, hm


class DetectionDecode:
    """Mirror of centernet/src/decode.py:123-196 (NMS :40-64, GatherTopK :90-109): feature dict with
    'hm' [B,C,H,W] (already sigmoid+clip), 'wh', 'reg' [B,2,H,W] fp32 NCHW -> detections [B,K,6]."""

    def __init__(self, reg_offset=True, K=100):
        self.reg_offset, self.K = reg_offset, K

    def __call__(self, feature, return_indices=False):
        wh = _f32c(feature["wh"])
        reg = _f32c(feature["reg"]) if self.reg_offset else None
        K = self.K
        if feature.get("heat") is not None:      # peaks already extracted (heat_peaks: the fused head post-processing)
            heat = _f32c(feature["heat"])
        else:
            hm = _f32c(feature["hm"])
            heat = torch.empty_like(hm)
            _lib.call("md_heat_nms", [hm, heat])
        B, C, H, W = heat.shape
        v1, i1 = top_k(heat.view(B * C, H * W), K)           # per-class top-K  (decode.py:96)
        v2, i2 = top_k(v1.view(B, C * K), K)                 # global top-K     (decode.py:101)
        det = torch.empty((B, K, 6), dtype=torch.float32, device=heat.device)
        inds = torch.empty((B, K), dtype=torch.int32, device=heat.device)
        cls = torch.empty((B, K), dtype=torch.int32, device=heat.device)
        _lib.call("md_centernet_assemble", [v2, i2, i1.view(B, C, K), wh, reg, det, inds, cls])
        return (det, inds, cls) if return_indices else det

    construct = __call__


# ----------------------------------------------------------------------------- two-stage glue
class _RpnDecodeAttrs(ctypes.Structure):
    _fields_ = [("num_anchors", ctypes.c_int32), ("decode", _DeltaAttrs)]


class _RcnnAttrs(ctypes.Structure):
    _fields_ = [("num_classes", ctypes.c_int32), ("reg_offset", ctypes.c_int32), ("score_thr", ctypes.c_float),
                ("decode", _DeltaAttrs)]


def _decode_attrs(means, stds, img_hw, wh_ratio_clip=16 / 1000):
    at = _DeltaAttrs()
    for i in range(4):
        at.means[i], at.stds[i] = float(means[i]), float(stds[i])
    at.max_ratio = abs(math.log(wh_ratio_clip))
    at.clip_h, at.clip_w = (float(img_hw[0]), float(img_hw[1])) if img_hw is not None else (0.0, 0.0)
    return at


def rpn_decode(head, anchors, idx, cnt, num_anchors, img_hw, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), out_boxes=None,
               out_scores=None):
    B, k = idx.shape
    if out_boxes is None:
        out_boxes = torch.empty((B, k, 4), dtype=torch.float32, device=head.device)
    if out_scores is None:
        out_scores = torch.empty((B, k), dtype=torch.float32, device=head.device)
    at = _RpnDecodeAttrs(int(num_anchors), _decode_attrs(means, stds, img_hw))
    _lib.call("md_rpn_decode", [head, anchors, idx, cnt, out_boxes, out_scores], extra=at)
    return out_boxes, out_scores


def rpn_merge(boxes, scores, keep):
    L, B, k = scores.shape
    mboxes = torch.empty((B, L * k, 4), dtype=torch.float32, device=boxes.device)
    mscores = torch.empty((B, L * k), dtype=torch.float32, device=boxes.device)
    _lib.call("md_rpn_merge", [boxes, scores, keep, mboxes, mscores])
    return mboxes, mscores


def make_rois(mboxes, topv, topi, cnt):
    B, post = topv.shape
    rois = torch.empty((B * post, 5), dtype=torch.float32, device=mboxes.device)
    rs = torch.empty((B * post,), dtype=torch.float32, device=mboxes.device)
    _lib.call("md_make_rois", [mboxes, topv, topi, cnt, rois, rs])
    return rois, rs


def rcnn_scores(cls_reg, roi_cnt, num_classes, score_thr):
    R = cls_reg.shape[0]
    B = roi_cnt.numel()
    cand = torch.empty((B, (R // B) * num_classes), dtype=torch.float32, device=cls_reg.device)
    at = _RcnnAttrs(int(num_classes), 0, float(score_thr), _DeltaAttrs())
    _lib.call("md_rcnn_scores", [cls_reg, roi_cnt, cand], extra=at)
    return cand


def rcnn_decode_selected(cls_reg, rois, sel_idx, sel_cnt, num_classes, reg_offset, img_hw, means=(0, 0, 0, 0),
                         stds=(0.1, 0.1, 0.2, 0.2)):
    B, npre = sel_idx.shape
    boxes = torch.empty((B, npre, 4), dtype=torch.float32, device=cls_reg.device)
    labels = torch.empty((B, npre), dtype=torch.int32, device=cls_reg.device)
    at = _RcnnAttrs(int(num_classes), int(reg_offset), 0.0, _decode_attrs(means, stds, img_hw))
    _lib.call("md_rcnn_decode_selected", [cls_reg, rois, sel_idx, sel_cnt, boxes, labels], extra=at)
    return boxes, labels


def pack_detections(boxes, scores, labels, keep_idx, num, max_det, sel_cnt=None, status=None):
    """-> dets [B,max_det,6], count [B].  With sel_cnt (the pre-NMS top-k's counts) and status ([B] int32, caller-cleared, in/out)
    bit 0 of status[b] is OR-ed in when the top-npre prefix was full AND gave fewer than max_det survivors: only then can the cut
    differ from the NMS over every candidate (see PrefixStatus)."""
    B = scores.shape[0]
    dets = torch.empty((B, max_det, 6), dtype=torch.float32, device=boxes.device)
    count = torch.empty((B,), dtype=torch.int32, device=boxes.device)
    if status is None:
        _lib.call("md_pack_detections", [boxes, scores, labels, keep_idx, num, dets, count])
    else:
        _lib.call("md_pack_detections", [boxes, scores, labels, keep_idx, num, sel_cnt, dets, count, status])
    return dets, count


class PrefixStatus:
    """Per-image sticky flags of the pre-NMS prefix cut (md_pack_detections, 9-parameter form), kept on the device: the hot path
    never synchronises for them; `flagged()` reads them (one host sync) when the caller wants to know, `clear()` resets."""

    def __init__(self):
        self._t = {}

    def tensor(self, batch, device):
        # one flag tensor per (batch, device, HIP stream): two streams running parts of a batch (graphs.SplitForward) never share a row
        key = (int(batch), str(device), torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0)
        if key not in self._t:
            self._t[key] = torch.zeros((batch,), dtype=torch.int32, device=device)
        return self._t[key]

    def flagged(self):
        """image slots (of any batch size seen) whose result MAY differ from the untruncated class-wise NMS."""
        return int(sum(int((t & 1).sum().item()) for t in self._t.values()))

    def clear(self):
        for t in self._t.values():
            t.zero_()


# ----------------------------------------------------------------------------- CenterPoint head post-processing
class _CenterPointAttrs(ctypes.Structure):
    _fields_ = [("off_reg", ctypes.c_int32), ("off_height", ctypes.c_int32), ("off_dim", ctypes.c_int32),
                ("off_rot", ctypes.c_int32), ("off_vel", ctypes.c_int32), ("off_hm", ctypes.c_int32),
                ("num_classes", ctypes.c_int32), ("score_threshold", ctypes.c_float), ("out_size_factor", ctypes.c_float),
                ("voxel_size", ctypes.c_float * 2), ("pc_range", ctypes.c_float * 2), ("post_center_range", ctypes.c_float * 6)]


def gather_rows(src, idx, cnt=None):
    """src [B,n,W] f32, idx [B,k] i32 -> [B,k,W] (zero rows past cnt)."""
    B, k = idx.shape
    out = torch.empty((B, k, src.shape[2]), dtype=torch.float32, device=src.device)
    _lib.call("md_gather_rows", [src, idx, cnt, out])
    return out


class CenterHeadPost:
    """Mirror of CenterHead.predict + post_processing for ONE task
    (centerpoint/det3d_ms/models/bbox_heads/center_head.py:273-463): decode every BEV cell, mask by score and
    post_center_range, TopK(nms_pre_max_size), rotated NMS through the AOT operator's device twin
    (`self.nms(boxes_for_nms_sorted, thr)`, :443-445), count = min(num_out, mask_num, nms_post_max_size)."""

    def __init__(self, offsets, num_classes, test_cfg):
        self.off, self.nc, self.cfg = dict(offsets), int(num_classes), test_cfg

    def __call__(self, head, return_aux=False):
        B, H, W, C = head.shape
        cfg = self.cfg
        at = _CenterPointAttrs()
        at.off_reg, at.off_height, at.off_dim, at.off_rot = self.off["reg"], self.off["height"], self.off["dim"], self.off["rot"]
        at.off_vel, at.off_hm, at.num_classes = self.off.get("vel", -1), self.off["hm"], self.nc
        at.score_threshold, at.out_size_factor = float(cfg["score_threshold"]), float(cfg["out_size_factor"])
        for i in range(2):
            at.voxel_size[i], at.pc_range[i] = float(cfg["voxel_size"][i]), float(cfg["pc_range"][i])
        for i in range(6):
            at.post_center_range[i] = float(cfg["post_center_limit_range"][i])
        dev = head.device
        n = H * W
        scores = torch.empty((B, n), dtype=torch.float32, device=dev)
        labels = torch.empty((B, n), dtype=torch.int32, device=dev)
        boxes = torch.empty((B, n, 9), dtype=torch.float32, device=dev)
        nms_boxes = torch.empty((B, n, 7), dtype=torch.float32, device=dev)
        _lib.call("md_centerpoint_decode", [head, scores, labels, boxes, nms_boxes], extra=at)
        k = int(cfg["nms"]["nms_pre_max_size"])
        sc_sorted, order = top_k(scores, k)                       # TopK over ALL cells, masked ones carry -1 (:435)
        nb_sorted = gather_rows(nms_boxes, order)
        bx_sorted = gather_rows(boxes, order)
        mask_num = (sc_sorted > -1.0).sum(1).to(torch.int32)      # == mask[order].sum() (:439-441)
        nms = NMS()
        keeps, nums = [], []
        for b in range(B):                                       # the reference loops the batch too (:405)
            kp, nm = nms(nb_sorted[b], cfg["nms"]["nms_iou_threshold"])
            keeps.append(kp)
            nums.append(nm.reshape(1))
        keep = torch.stack(keeps)
        num_out = torch.cat(nums)
        count = torch.minimum(torch.minimum(num_out, mask_num), torch.full_like(num_out, int(cfg["nms"]["nms_post_max_size"])))
        sel_boxes = gather_rows(bx_sorted, keep)
        sel_scores = torch.gather(sc_sorted, 1, keep.long())
        sel_labels = torch.gather(torch.gather(labels, 1, order.long()), 1, keep.long())
        out = [sel_boxes, sel_scores, sel_labels, count]
        if return_aux:
            return out, dict(scores=scores, labels=labels, boxes=boxes, nms_boxes=nms_boxes, order=order, keep=keep,
                             num_out=num_out, mask_num=mask_num)
        return out


# ----------------------------------------------------------------------------- PointPillars host post-process
def _just_below(x):
    """Largest float32 strictly below x: `score >= x` (predict.py:30) expressed as `score > just_below(x)`."""
    import numpy as np

    return float(np.nextafter(np.float32(x), np.float32(-np.inf)))


def standup_boxes(boxes):
    """Rotated BEV boxes [N,5] (x,y,dx,dy,r) or [N,7] -> standup boxes [N,4] (predict.py:61-78)."""
    b = _f32c(boxes)
    out = torch.empty((b.shape[0], 4), dtype=torch.float32, device=b.device)
    _lib.call("md_standup_boxes", [b, out])
    return out


def pp_get_selected_data(total_scores, box_preds, anchors_mask, cfg):
    """Mirror of PointPillarsNet.get_selected_data (pointpillars/src/pointpillars.py:753-765) followed by the host
    post-process _get_selected_data (pointpillars/src/predict.py:43-98) for one sample, on device:
    class max / argmax, anchor-mask -> -1, top_k(nms_pre_max_size), score threshold, standup boxes, NMS on the
    standup boxes (nms_jit convention: the reference's ops.NMSWithMask arithmetic is not in the repository),
    first nms_post_max_size survivors.  Returns (boxes[K,7], scores[K], labels[K], count)."""
    top_scores, top_labels = total_scores.max(-1)
    top_scores = torch.where(anchors_mask, top_scores, torch.full_like(top_scores, -1.0))
    k = min(int(cfg["nms_pre_max_size"]), top_scores.numel())
    vals, idx, cnt = topk_segmented(top_scores, torch.tensor([0, top_scores.numel()], dtype=torch.int32, device=top_scores.device),
                                    k, min_score=_just_below(cfg["nms_score_threshold"]) if cfg["nms_score_threshold"] > 0 else None,
                                    max_segment=top_scores.numel())
    sel = gather_rows(_f32c(box_preds).unsqueeze(0), idx, cnt)[0]      # [k,7], zero rows past cnt
    st = standup_boxes(sel)
    mask, kidx, num = nms_aligned(st.unsqueeze(0), float(cfg["nms_iou_threshold"]), 0.0, NMS_MODE_JIT, count=cnt,
                                  max_output=int(cfg["nms_post_max_size"]))
    n = num[0]
    ki = kidx[0].long()
    return sel[ki], vals[0][ki], top_labels[idx[0].long()][ki], n


# ----------------------------------------------------------------------------- CenterNet post-process (post_process.py)
class _SoftNmsAttrs(ctypes.Structure):
    _fields_ = [("sigma", ctypes.c_float), ("Nt", ctypes.c_float), ("threshold", ctypes.c_float), ("method", ctypes.c_int32)]


def soft_nms(boxes, scores, count=None, sigma=0.5, Nt=0.5, threshold=0.001, method=2):
    """boxes [L,N,4] / [N,4], scores [L,N] / [N] -> (scores_out, order, num). See md_soft_nms."""
    b, s = _f32c(boxes), _f32c(scores)
    batched = b.dim() == 3
    L, n = (b.shape[0], b.shape[1]) if batched else (1, b.shape[0])
    so = torch.empty((L, n), dtype=torch.float32, device=b.device)
    order = torch.empty((L, n), dtype=torch.int32, device=b.device)
    num = torch.empty((L,), dtype=torch.int32, device=b.device)
    _lib.call("md_soft_nms", [b.view(L, n, 4), s.view(L, n), count, so, order, num], extra=_SoftNmsAttrs(sigma, Nt, threshold, method))
    return (so, order, num) if batched else (so[0], order[0], num)


def get_affine_transform(center, scale, output_size, inv=True):
    """centernet/src/image.py (get_affine_transform with rot = 0): the 2x3 matrix cv2.getAffineTransform returns for
    the three reference points, solved here in float64 (cv2 is not a dependency)."""
    import numpy as np

    scale = np.array([scale, scale], np.float32) if np.isscalar(scale) else np.asarray(scale, np.float32)
    src_w, dst_w, dst_h = scale[0], output_size[0], output_size[1]
    center = np.asarray(center, np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0] = center
    src[1] = center + np.array([0, src_w * -0.5], np.float32)
    dst[0] = [dst_w * 0.5, dst_h * 0.5]
    dst[1] = np.array([dst_w * 0.5, dst_h * 0.5], np.float32) + np.array([0, dst_w * -0.5], np.float32)
    for p in (src, dst):
        d = p[0] - p[1]
        p[2] = p[1] + np.array([-d[1], d[0]], np.float32)
    a, b = (dst, src) if inv else (src, dst)
    A = np.concatenate([a.astype(np.float64), np.ones((3, 1))], 1)
    return np.linalg.solve(A, b.astype(np.float64)).T  # [2,3]


def centernet_post_process(dets, c, s, out_hw, scale, num_classes, soft=True, max_per_image=100):
    """Mirror of post_process + merge_outputs (centernet/src/post_process.py:10-61) for one image, on device.
    dets [K,6] (x1,y1,x2,y2,score,cls) in feature-map units -> rows surviving the per-class (soft-)NMS and the
    global top-max_per_image cut, in image coordinates: (boxes [M,4], scores [M], classes [M])."""
    t = torch.from_numpy(get_affine_transform(c, s, (out_hw[1], out_hw[0]))).to(dets.device)
    d = dets.to(torch.float64)
    xy1 = (d[:, 0:2] @ t[:, :2].T + t[:, 2]).to(torch.float32) / scale
    xy2 = (d[:, 2:4] @ t[:, :2].T + t[:, 2]).to(torch.float32) / scale
    boxes = torch.cat([xy1, xy2], 1).contiguous()
    scores, cls = dets[:, 4].contiguous(), dets[:, 5].to(torch.int32)
    K = boxes.shape[0]
    if soft:
        # per-class lists: class-major padding [C, K]
        lists_b = torch.zeros((num_classes, K, 4), dtype=torch.float32, device=dets.device)
        lists_s = torch.zeros((num_classes, K), dtype=torch.float32, device=dets.device)
        cnt = torch.zeros((num_classes,), dtype=torch.int32, device=dets.device)
        pos = torch.zeros((K,), dtype=torch.long, device=dets.device)
        for j in range(num_classes):  # classes == j split (post_process.py:19-27); tiny host loop over <= 80 classes
            idx = torch.nonzero(cls == j).flatten()
            m = idx.numel()
            if m:
                lists_b[j, :m], lists_s[j, :m], cnt[j] = boxes[idx], scores[idx], m
                pos[idx] = torch.arange(m, device=dets.device)
        so, _, _ = soft_nms(lists_b, lists_s, cnt)
        scores = so[cls.long(), pos]
        alive = scores > 0
    else:
        alive = torch.ones_like(scores, dtype=torch.bool)
    s_alive = scores[alive]
    if s_alive.numel() > max_per_image:  # np.partition threshold, ties may exceed max_per_image (post_process.py:54-60)
        thresh = torch.sort(s_alive)[0][s_alive.numel() - max_per_image]
        alive = alive & (scores >= thresh)
    return boxes[alive], scores[alive], cls[alive]


class _RotIouAttrs(ctypes.Structure):
    _fields_ = [("criterion", ctypes.c_int32)]


def rotate_iou_gpu_eval(boxes, query_boxes, criterion=-1):
    """pointpillars/eval_gpu/rotate_iou.py:305-340: [N,5] x [K,5] (cx,cy,dx,dy,angle) -> [N,K]."""
    b, q = _f32c(boxes), _f32c(query_boxes)
    out = torch.empty((b.shape[0], q.shape[0]), dtype=torch.float32, device=b.device)
    _lib.call("md_rotate_iou_eval", [b, q, out], extra=_RotIouAttrs(int(criterion)))
    return out


# ----------------------------------------------------------------------------- YOLOv5 Detect decode
class _YoloAttrs(ctypes.Structure):
    _fields_ = [("num_classes", ctypes.c_int32), ("num_anchors", ctypes.c_int32), ("stride", ctypes.c_float),
                ("anchors", ctypes.c_float * 6), ("conf_thres", ctypes.c_float), ("out_offset", ctypes.c_int32),
                ("out_total", ctypes.c_int32)]


def yolo_decode(head, boxes, scores, labels, num_classes, num_anchors, stride, anchors, conf_thres, out_offset, out_total):
    at = _YoloAttrs(int(num_classes), int(num_anchors), float(stride))
    for i, v in enumerate(anchors):
        at.anchors[i] = float(v)
    at.conf_thres, at.out_offset, at.out_total = float(conf_thres), int(out_offset), int(out_total)
    _lib.call("md_yolo_decode", [head, boxes, scores, labels], extra=at)


def assign_targets(anchors, gt_boxes, gt_classes, matched_thr, unmatched_thr, anchors_mask=None):
    """create_target_np (pointpillars/src/core/target_assigner.py:29-166; TargetAssigner.assign :196-224) on the device:
    -> (labels [A] i32, bbox_targets [A,7] f32, bbox_outside_weights [A] f32, gt_ids [A] i32).  Thresholds: float or [A]."""
    a = _f32c(anchors).reshape(-1, 7)
    dev, A = a.device, a.shape[0]
    g = _f32c(gt_boxes).reshape(-1, 7)
    G = g.shape[0]
    cls = (torch.ones((G,), dtype=torch.int32, device=dev) if gt_classes is None
           else gt_classes.to(device=dev, dtype=torch.int32).contiguous())

    def thr(v):
        return (torch.full((A,), float(v), dtype=torch.float32, device=dev) if not torch.is_tensor(v) else _f32c(v).reshape(-1))

    mt, ut = thr(matched_thr), thr(unmatched_thr)
    mask = None if anchors_mask is None else anchors_mask.to(device=dev, dtype=torch.uint8).contiguous()
    labels = torch.empty((A,), dtype=torch.int32, device=dev)
    targets = torch.empty((A, 7), dtype=torch.float32, device=dev)
    weights = torch.empty((A,), dtype=torch.float32, device=dev)
    gt_ids = torch.empty((A,), dtype=torch.int32, device=dev)
    _lib.call("md_assign_targets", [a, g if G else None, cls if G else None, mt, ut, mask, labels, targets, weights, gt_ids])
    return labels, targets, weights, gt_ids


def mask_select(logits, dets, num_classes):
    """[R,S,S,Cpad] bf16 mask logits + [R,6] detections -> [R,S,S] f32 sigmoid of each detection's own class channel."""
    r, sz = logits.shape[0], logits.shape[1]
    out = torch.empty((r, sz, sz), dtype=torch.float32, device=logits.device)
    _lib.call("md_mask_select", [logits, _f32c(dets).reshape(-1, 6), out], extra=ctypes.c_int32(int(num_classes)))
    return out


class _PasteAttrs(ctypes.Structure):
    _fields_ = [("img_h", ctypes.c_int32), ("img_w", ctypes.c_int32), ("threshold", ctypes.c_float), ("bits", ctypes.c_int32)]


def paste_masks(masks, dets, img_hw, threshold=0.5, bits=True):
    """[B,D,S,S] f32 mask probabilities + [B,D,6] detections -> image-resolution masks (md_paste_masks): bits=True ->
    [B,D,H,ceil(W/32)] int32 words (bit j of word k = pixel 32 k + j), else [B,D,H,W] uint8 (0 / 1)."""
    B, D, S = masks.shape[0], masks.shape[1], masks.shape[2]
    H, W = int(img_hw[0]), int(img_hw[1])
    if bits:
        out = torch.empty((B * D, H, (W + 31) // 32), dtype=torch.int32, device=masks.device)
    else:
        out = torch.empty((B * D, H, W), dtype=torch.uint8, device=masks.device)
    _lib.call("md_paste_masks", [_f32c(masks).reshape(B * D, S, S), _f32c(dets).reshape(B * D, 6), out],
              extra=_PasteAttrs(H, W, float(threshold), int(bool(bits))))
    return out.view(B, D, H, out.shape[2])


def unpack_mask_bits(words, width):
    """[..., H, ceil(W/32)] int32 bit masks -> [..., H, W] uint8 (host-side helper for tests / result export)."""
    w = words.to(torch.int64) & 0xffffffff
    sh = torch.arange(32, device=words.device, dtype=torch.int64)
    return ((w.unsqueeze(-1) >> sh) & 1).reshape(*words.shape[:-1], -1)[..., :width].to(torch.uint8)


def dets_to_rois(dets):
    """[B,D,6] detections -> [B*D,5] RoIs (batch index, x1, y1, x2, y2) for the mask head (layout plumbing)."""
    B, D = dets.shape[0], dets.shape[1]
    b = torch.arange(B, dtype=torch.float32, device=dets.device).view(B, 1, 1).expand(B, D, 1)
    return torch.cat([b, dets[..., :4]], -1).reshape(B * D, 5).contiguous()


class _Yolo8Attrs(ctypes.Structure):
    _fields_ = [("num_classes", ctypes.c_int32), ("reg_max", ctypes.c_int32), ("stride", ctypes.c_float), ("conf_thres", ctypes.c_float),
                ("out_offset", ctypes.c_int32), ("out_total", ctypes.c_int32)]


def yolov8_decode(head, boxes, scores, labels, num_classes, reg_max, stride, conf_thres, out_offset, out_total):
    _lib.call("md_yolov8_decode", [head, boxes, scores, labels],
              extra=_Yolo8Attrs(int(num_classes), int(reg_max), float(stride), float(conf_thres), int(out_offset), int(out_total)))
