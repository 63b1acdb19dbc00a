"""CPU oracle -- TEST INFRASTRUCTURE ONLY.

Restates the reference's detection-op algorithms on the CPU (plain C in det_oracle.c,
numpy/torch-CPU in np_ops.py / nets.py).  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import this package; the product package
``minddet_amd`` never does (tests/test_abi_cpu.py::test_product_never_imports_oracle enforces it).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None

_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "det_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, os.path.join(_HERE, "liboracle.so")])
    if os.path.isdir("/root/reference") and (
        force or not os.path.exists(os.path.join(_HERE, "_ref", "libref_nms.so"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.orc_rot_overlap.restype = ctypes.c_float
        _LIB.orc_iou_bev.restype = ctypes.c_float
    return _LIB


def ref_lib():
    """The reference's own iou-bev-nms-org.cpp compiled by oracle/Makefile (None if absent)."""
    global _REF
    if _REF is None:
        p = os.path.join(_HERE, "_ref", "libref_nms.so")
        if not os.path.exists(p):
            if os.path.isdir("/root/reference"):
                build()
            if not os.path.exists(p):
                return None
        _REF = ctypes.CDLL(p)
    return _REF


def _c(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ---------------------------------------------------------------- C oracle wrappers
def boxes_iou_bev(a, b):
    a, b = _c(a, np.float32), _c(b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_iou_bev(_p(a), ctypes.c_int64(a.shape[0]), _p(b), ctypes.c_int64(b.shape[0]), _p(out))
    return out


def boxes_overlap_bev(a, b):
    a, b = _c(a, np.float32), _c(b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    lib().orc_boxes_overlap_bev(_p(a), ctypes.c_int64(a.shape[0]), _p(b), ctypes.c_int64(b.shape[0]), _p(out))
    return out


def nms_rot_aot(boxes, thr):
    """boxes_iou_nms_cpu semantics -> (keep[N] int32 zero-padded, num)."""
    boxes = _c(boxes, np.float32)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int32)
    num = np.zeros(1, np.int32)
    lib().orc_nms_rot_aot(_p(boxes), ctypes.c_int64(n), ctypes.c_float(thr), _p(keep), _p(num))
    return keep[:n], int(num[0])


def nms_rot_mask(boxes, thr):
    boxes = _c(boxes, np.float32)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    num = np.zeros(1, np.int32)
    lib().orc_nms_rot_mask(_p(boxes), ctypes.c_int64(n), ctypes.c_float(thr), _p(keep), _p(num))
    return keep[:n], int(num[0])


def nms_normal_mask(boxes, thr):
    boxes = _c(boxes, np.float32)
    n = boxes.shape[0]
    keep = np.zeros(max(n, 1), np.int64)
    num = np.zeros(1, np.int32)
    lib().orc_nms_normal_mask(_p(boxes), ctypes.c_int64(n), ctypes.c_float(thr), _p(keep), _p(num))
    return keep[:n], int(num[0])


def iou_aligned(boxes, query, eps=0.0):
    boxes, query = _c(boxes, np.float32), _c(query, np.float32)
    out = np.zeros((boxes.shape[0], query.shape[0]), np.float32)
    lib().orc_iou_aligned(_p(boxes), ctypes.c_int64(boxes.shape[0]), _p(query),
                          ctypes.c_int64(query.shape[0]), ctypes.c_float(eps), _p(out))
    return out


def nms_aligned(boxes_sorted, thr, eps=0.0, mode=0, groups=None):
    """Greedy NMS on score-sorted corner boxes -> uint8 keep-mask. See det_oracle.c."""
    b = _c(boxes_sorted, np.float32)
    n = b.shape[0]
    mask = np.zeros(max(n, 1), np.uint8)
    g = None if groups is None else _c(groups, np.int32)
    lib().orc_nms_aligned(_p(b), None if g is None else _p(g), ctypes.c_int64(n), ctypes.c_float(thr),
                          ctypes.c_float(eps), ctypes.c_int(mode), _p(mask))
    return mask[:n]


def circle_nms(dets_sorted, thresh):
    d = _c(dets_sorted, np.float32)
    n = d.shape[0]
    mask = np.zeros(max(n, 1), np.uint8)
    lib().orc_circle_nms(_p(d), ctypes.c_int64(n), ctypes.c_float(thresh), _p(mask))
    return mask[:n]


def rotate_iou_eval(boxes, query, criterion=-1):
    """pointpillars/eval_gpu/rotate_iou.py:305-340 semantics: [N,5] x [K,5] -> [N,K]."""
    b, q = _c(boxes, np.float32), _c(query, np.float32)
    out = np.zeros((b.shape[0], q.shape[0]), np.float32)
    lib().orc_rotate_iou_eval(_p(b), ctypes.c_int64(b.shape[0]), _p(q), ctypes.c_int64(q.shape[0]), ctypes.c_int(criterion), _p(out))
    return out


# ---------------------------------------------------------------- reference (oracle/_ref) wrappers
def ref_boxes_iou_nms_cpu(boxes1000, thr):
    """Drive the reference operator through its AOT ABI. N is hard-coded to 1000 inside it."""
    r = ref_lib()
    assert r is not None, "oracle/_ref not built"
    boxes = _c(boxes1000, np.float32)
    assert boxes.shape == (1000, 7)
    t = np.array([thr], np.float32)
    keep = np.zeros(1000, np.int32)
    num = np.zeros(1, np.int32)
    params = (ctypes.c_void_p * 4)(boxes.ctypes.data, t.ctypes.data, keep.ctypes.data, num.ctypes.data)
    rc = r.boxes_iou_nms_cpu(4, params, None, None, None, None, None)
    assert rc == 0
    return keep, int(num[0])


def ref_boxes_iou_bev_cpu(a, b):
    r = ref_lib()
    assert r is not None, "oracle/_ref not built"
    a, b = _c(a, np.float32), _c(b, np.float32)
    out = np.zeros((a.shape[0], b.shape[0]), np.float32)
    fn = getattr(r, "_Z17boxes_iou_bev_cpuPKfiS0_iPf")
    fn(_p(a), ctypes.c_int(a.shape[0]), _p(b), ctypes.c_int(b.shape[0]), _p(out))
    return out
