"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Run once in the build container (needs /root/reference; never on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/gen_golden.py

Sources of truth:
  * oracle/_ref/libref_nms.so -- the reference's own
    minddet/models/centerpoint/det3d_ms/ops/iou-bev-nms-org.cpp compiled by oracle/Makefile
    (boxes_iou_nms_cpu through its AOT ABI, boxes_iou_bev_cpu).
  * the reference's numpy functions in minddet/models/pointpillars/src/core/
    {box_np_ops,nms}.py imported under an import-time shim (numba.jit -> identity,
    mindspore -> mock; ms.ops.meshgrid -> np.meshgrid(indexing="ij"), the one MindSpore call
    inside create_anchors_3d_stride).  Nothing of the reference is copied: the fixtures hold
    only seeded inputs and the outputs the reference produced for them.
"""
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def _shim():
    nb = types.ModuleType("numba")

    def _ident(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f

    nb.jit = _ident
    nb.njit = _ident
    sys.modules["numba"] = nb
    ms = MagicMock()

    class _T:
        def __init__(self, a):
            self.a = a

        def asnumpy(self):
            return self.a

    ms.Tensor.from_numpy = lambda a: a
    ms.ops.meshgrid = lambda *xs, indexing="ij": tuple(_T(g) for g in np.meshgrid(*xs, indexing=indexing))
    for m in ["mindspore", "mindspore.ops", "mindspore.nn", "mindspore.common", "mindspore.common.dtype"]:
        sys.modules[m] = ms if m == "mindspore" else getattr(ms, m.split(".", 1)[1].replace(".", "_"), MagicMock())
    sys.modules["mindspore.ops"] = ms.ops
    sys.path.insert(0, "/root/reference/minddet/models/pointpillars")
    return _T


def rot_boxes(n, rng, span):
    b = np.zeros((n, 7), np.float32)
    b[:, 0:2] = rng.uniform(-span, span, (n, 2))
    b[:, 2] = rng.uniform(-2, 2, n)
    b[:, 3:6] = rng.uniform(1, 5, (n, 3))
    b[:, 6] = rng.uniform(-np.pi, np.pi, n)
    return b


def aligned_boxes(n, rng, W=1344.0, H=800.0):
    cx = rng.uniform(0, W, n)
    cy = rng.uniform(0, H, n)
    w = np.exp(rng.uniform(np.log(8), np.log(512), n))
    h = np.exp(rng.uniform(np.log(8), np.log(512), n))
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], -1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, W)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, H)
    return b.astype(np.float32)


def main():
    import oracle

    T = _shim()
    from src.core import box_np_ops, nms as ref_nms

    out = {}
    # ---- rotated NMS through the reference AOT operator (N hard-coded 1000 inside it)
    rng = np.random.default_rng(0)
    for tag, span, thr in [("a", 50.0, 0.2), ("b", 12.0, 0.2), ("c", 6.0, 0.5), ("d", 25.0, 0.01)]:
        b = rot_boxes(1000, rng, span)
        if tag == "c":  # exercise the zero-area pre-removal (iou-bev-nms-org.cpp:253-255)
            b[::97, 3] = 0
        keep, num = oracle.ref_boxes_iou_nms_cpu(b, thr)
        out[f"rotnms_{tag}_boxes"] = b
        out[f"rotnms_{tag}_thr"] = np.float32(thr)
        out[f"rotnms_{tag}_keep"] = keep
        out[f"rotnms_{tag}_num"] = np.int32(num)
    # ---- rotated IoU matrix
    a = rot_boxes(160, rng, 8.0)
    c = rot_boxes(24, rng, 8.0)
    a[5] = c[3]  # identical pair -> IoU 1 path (corner-in-box only)
    out["ioubev_a"], out["ioubev_b"] = a, c
    out["ioubev_out"] = oracle.ref_boxes_iou_bev_cpu(a, c)
    # ---- axis-aligned IoU (iou_jit)
    rng = np.random.default_rng(11)
    bx = aligned_boxes(300, rng)
    qx = aligned_boxes(20, rng)
    out["ioujit_boxes"], out["ioujit_query"] = bx, qx
    out["ioujit_out_eps0"] = box_np_ops.iou_jit(bx, qx, 0.0)
    out["ioujit_out_eps1"] = box_np_ops.iou_jit(bx, qx, 1.0)
    # ---- nms_jit
    dets = np.concatenate([aligned_boxes(400, rng, 400, 300), np.zeros((400, 1), np.float32)], 1)
    s = 1 / (1 + np.exp(-rng.normal(-3, 2, 400)))
    dets[:, 4] = (s + np.arange(400) * 1e-7).astype(np.float32)
    out["nmsjit_dets"] = dets
    for thr in (0.01, 0.5, 0.7):
        out[f"nmsjit_keep_{thr}"] = np.array(ref_nms.nms_jit(dets, thr, 0.0), np.int32)
    out["nmsjit_keep_0.5_eps1"] = np.array(ref_nms.nms_jit(dets, 0.5, 1.0), np.int32)
    # ---- apply_nms (+1 convention); boxes are (y1,x1,y2,x2) there
    yx = dets[:, [1, 0, 3, 2]].copy()
    out["applynms_keep"] = ref_nms.apply_nms(T(yx), T(dets[:, 4].copy()), 0.5, 100).astype(np.int32)
    # ---- near bbox / limit_period / corners
    rb = np.concatenate([rng.uniform(-40, 40, (200, 2)), rng.uniform(0.5, 5, (200, 2)),
                         rng.uniform(-4, 4, (200, 1))], 1).astype(np.float32)
    out["near_in"] = rb
    out["near_out"] = box_np_ops.rbbox2d_to_near_bbox(rb)
    out["corners_out"] = box_np_ops.center_to_corner_box2d(rb[:, :2], rb[:, 2:4], rb[:, 4])
    out["standup_out"] = box_np_ops.corner_to_standup_nd(out["corners_out"])
    # ---- codec
    anc = np.concatenate([rng.uniform(-40, 40, (256, 3)), rng.uniform(0.5, 5, (256, 3)),
                          rng.uniform(-3, 3, (256, 1))], 1).astype(np.float32)
    gt = (anc + rng.normal(0, 0.3, anc.shape)).astype(np.float32)
    gt[:, 3:6] = np.abs(gt[:, 3:6]) + 0.1
    enc = box_np_ops.second_box_encode(gt, anc)
    out["codec_anchors"], out["codec_boxes"], out["codec_enc"] = anc, gt, enc
    out["codec_dec"] = box_np_ops.second_box_decode(enc, anc)
    # ---- anchors (PP car config: configs/car_xyres16.yaml:119-127)
    fs = [1, 31, 27]
    car = dict(sizes=[1.6, 3.9, 1.56], anchor_strides=[0.32, 0.32, 0.0], anchor_offsets=[0.16, -39.52, -1.78],
               rotations=[0, 1.57], anchor_range=[0, -39.68, -3, 69.12, 39.68, 1])
    out["anchors_stride_small"] = box_np_ops.create_anchors_3d_stride(fs, car["sizes"], car["anchor_strides"],
                                                                       car["anchor_offsets"], car["rotations"],
                                                                       car["anchor_range"], np.float32)
    full = box_np_ops.create_anchors_3d_stride([1, 248, 216], car["sizes"], car["anchor_strides"],
                                               car["anchor_offsets"], car["rotations"], car["anchor_range"],
                                               np.float32)
    # full map is 3 MB: keep a strided sample + a checksum instead
    flat = full.reshape(-1, 7)
    out["anchors_stride_full_shape"] = np.array(full.shape, np.int32)
    out["anchors_stride_full_sample_idx"] = np.arange(0, flat.shape[0], 997, dtype=np.int32)
    out["anchors_stride_full_sample"] = flat[::997]
    out["anchors_stride_full_sum64"] = flat.astype(np.float64).sum(0)
    # ---- anchor mask
    anchors_bv = box_np_ops.rbbox2d_to_near_bbox(flat[:, [0, 1, 3, 4, 6]])
    grid = np.array([432, 496, 1])
    coors = np.stack([np.zeros(6000, np.int64), rng.integers(0, 496, 6000), rng.integers(0, 432, 6000)], 1).astype(np.int32)
    dense = box_np_ops.sparse_sum_for_anchors_mask(coors, tuple(grid[::-1][1:]))
    dense = dense.cumsum(0).cumsum(1)
    vs = np.array([0.16, 0.16, 4.0], np.float32)
    pcr = np.array([0, -39.68, -3, 69.12, 39.68, 1], np.float32)
    sel = np.arange(0, flat.shape[0], 53)
    out["amask_coors"] = coors
    out["amask_sel"] = sel.astype(np.int32)
    out["amask_anchors_bv"] = anchors_bv[sel]
    out["amask_area"] = box_np_ops.fused_get_anchors_area(dense, anchors_bv[sel], vs, pcr, grid)

    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    tot = sum(v.nbytes for v in out.values())
    print("wrote", len(out), "arrays,", tot, "bytes raw ->",
          os.path.getsize(os.path.join(HERE, "reference_vectors.npz")), "bytes on disk")


if __name__ == "__main__":
    main()
