"""Golden vectors for target assignment, produced BY THE REFERENCE: create_target_np
(minddet/models/pointpillars/src/core/target_assigner.py:29-166) driven as TargetAssigner.assign drives it (:196-224):
similarity = NearestIouSimilarity (region_similarity.py:46-59: rbbox2d_to_near_bbox + iou_jit(eps=0)), encoding =
second_box_encode (box_np_ops.py:8-37), positive_fraction None (configs/car_xyres16.yaml:129).  Run here only:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/gen_targets.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from gen_golden import _shim  # noqa: E402


def main():
    _shim()
    from src.core import box_np_ops, target_assigner  # the reference's own modules

    def similarity(anchors, gt):
        a = box_np_ops.rbbox2d_to_near_bbox(anchors[:, [0, 1, 3, 4, 6]])
        g = box_np_ops.rbbox2d_to_near_bbox(gt[:, [0, 1, 3, 4, 6]])
        return box_np_ops.iou_jit(a, g, eps=0.0)

    def encode(boxes, anchors):
        return box_np_ops.second_box_encode(boxes, anchors)

    rng = np.random.default_rng(29)
    out = {}

    def anchors_grid(nx, ny, sizes, rots):
        xs = np.linspace(0.0, 69.12, nx, dtype=np.float32)
        ys = np.linspace(-39.68, 39.68, ny, dtype=np.float32)
        a = []
        for y in ys:
            for x in xs:
                for s in sizes:
                    for r in rots:
                        a.append([x, y, -1.0, s[0], s[1], s[2], r])
        return np.array(a, np.float32)

    def gts(n, span=1.0):
        g = np.zeros((n, 7), np.float32)
        g[:, 0] = rng.uniform(0, 69.12, n)
        g[:, 1] = rng.uniform(-39.68, 39.68, n)
        g[:, 2] = rng.uniform(-2, 0, n)
        g[:, 3] = rng.uniform(1.4, 1.9, n) * span
        g[:, 4] = rng.uniform(3.2, 4.6, n) * span
        g[:, 5] = rng.uniform(1.3, 1.8, n)
        g[:, 6] = rng.uniform(-np.pi, np.pi, n)
        return g

    cases = {
        "car": dict(anchors=anchors_grid(54, 62, [(1.6, 3.9, 1.56)], [0.0, 1.57]), gt=gts(17), mask=True, thr=(0.6, 0.45), cls=None),
        "nomask": dict(anchors=anchors_grid(40, 36, [(1.6, 3.9, 1.56)], [0.0, 1.57]), gt=gts(9), mask=False, thr=(0.6, 0.45), cls=None),
        "nogt": dict(anchors=anchors_grid(10, 12, [(1.6, 3.9, 1.56)], [0.0, 1.57]), gt=np.zeros((0, 7), np.float32), mask=True, thr=(0.6, 0.45), cls=None),
        "pedcyc": dict(anchors=anchors_grid(48, 50, [(0.6, 0.8, 1.73), (0.6, 1.76, 1.73)], [0.0, 1.57]), gt=gts(14, 0.4), mask=True,
                       thr="per_anchor", cls=rng.integers(1, 3, 14).astype(np.int32)),
    }
    # one ground-truth box that overlaps nothing (empty_gt_mask path) and one exactly on an anchor (IoU 1, ties)
    cases["car"]["gt"][3, 0:2] = [500.0, 500.0]
    cases["car"]["gt"][5] = cases["car"]["anchors"][1234]
    for name, c in cases.items():
        A = c["anchors"].shape[0]
        mask = (rng.uniform(0, 1, A) < 0.7) if c["mask"] else None
        if c["thr"] == "per_anchor":
            mt = np.where(np.arange(A) % 4 < 2, 0.5, 0.5).astype(np.float32)
            ut = np.where(np.arange(A) % 4 < 2, 0.35, 0.35).astype(np.float32)
            mt[::3] = 0.45
        else:
            mt, ut = c["thr"]
        r = target_assigner.create_target_np(c["anchors"], c["gt"], similarity, encode, gt_classes=c["cls"], matched_threshold=mt,
                                             unmatched_threshold=ut, positive_fraction=None, rpn_batch_size=512,
                                             norm_by_num_examples=False, box_code_size=7, anchors_mask=mask)
        out[name + "_anchors"] = c["anchors"]
        out[name + "_gt"] = c["gt"]
        out[name + "_mask"] = np.zeros((0,), np.uint8) if mask is None else mask.astype(np.uint8)
        out[name + "_mt"] = np.full((A,), mt, np.float32) if np.isscalar(mt) else mt
        out[name + "_ut"] = np.full((A,), ut, np.float32) if np.isscalar(ut) else ut
        out[name + "_cls"] = np.ones((c["gt"].shape[0],), np.int32) if c["cls"] is None else c["cls"]
        out[name + "_labels"] = r["labels"].astype(np.int32)
        out[name + "_targets"] = r["bbox_targets"].astype(np.float32)
        out[name + "_weights"] = r["bbox_outside_weights"].astype(np.float32)
        print(name, "anchors", A, "gt", c["gt"].shape[0], "pos", int((r["labels"] > 0).sum()), "neg", int((r["labels"] == 0).sum()))
    np.savez_compressed(os.path.join(HERE, "target_vectors.npz"), **out)


if __name__ == "__main__":
    main()
