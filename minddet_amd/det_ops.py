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
