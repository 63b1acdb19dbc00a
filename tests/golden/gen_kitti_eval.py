"""Golden vectors for the KITTI image-bbox AP protocol, produced BY THE REFERENCE: get_official_eval_result / eval_class /
compute_statistics_jit / clean_data / get_thresholds of minddet/models/pointpillars/src/core/eval_utils.py (the evaluator
pointpillars/eval.py:149 calls), imported under the numba -> identity shim of gen_golden.py (the functions are plain numpy /
Python loops).  Inputs are synthetic annotations; only inputs and outputs are stored.  Run here only:

    PYTHONDONTWRITEBYTECODE=1 python -B tests/golden/gen_kitti_eval.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
from gen_golden import _shim  # noqa: E402

NAMES = ["Car", "Pedestrian", "Cyclist", "Van", "Person_sitting", "DontCare"]


def make_annos(rng, n_img):
    gts, dts = [], []
    for _ in range(n_img):
        n = int(rng.integers(0, 9))
        names = rng.choice(NAMES, n, p=[0.4, 0.2, 0.12, 0.1, 0.06, 0.12])
        x1 = rng.uniform(0, 1100, n)
        y1 = rng.uniform(100, 250, n)
        h = rng.choice([18, 24, 30, 38, 45, 60, 90, 140], n).astype(float) + rng.uniform(0, 4, n)
        w = h * rng.uniform(0.4, 2.2, n)
        bbox = np.stack([x1, y1, x1 + w, y1 + h], 1)
        gt = dict(name=[str(s) for s in names], bbox=bbox, occluded=rng.integers(0, 4, n), truncated=rng.choice([0.0, 0.1, 0.2, 0.4, 0.6], n),
                  alpha=rng.uniform(-3, 3, n))
        dc = np.array([s == "DontCare" for s in gt["name"]], bool)
        gt["occluded"] = np.where(dc, -1, gt["occluded"])
        gt["truncated"] = np.where(dc, -1.0, gt["truncated"])
        # detections: most ground truth jittered (some heavily), plus strays; classes sometimes confused
        keep = rng.random(n) < 0.8
        dn = [rng.choice(NAMES[:3]) if (s == "DontCare" or rng.random() < 0.1) else s for s in np.array(gt["name"])[keep]]
        jit = rng.normal(0, 1, (keep.sum(), 4)) * rng.choice([0.5, 3.0, 12.0], (keep.sum(), 1), p=[0.6, 0.25, 0.15])
        db = bbox[keep] + jit
        m = int(rng.integers(0, 4))
        sx, sy = rng.uniform(0, 1100, m), rng.uniform(100, 250, m)
        sh = rng.uniform(15, 120, m)
        stray = np.stack([sx, sy, sx + sh * rng.uniform(0.5, 2, m), sy + sh], 1).reshape(m, 4)
        dbox = np.concatenate([db.reshape(-1, 4), stray], 0)
        dnames = [str(s) for s in dn] + [str(rng.choice(NAMES[:3])) for _ in range(m)]
        dt = dict(name=dnames, bbox=dbox, alpha=rng.uniform(-3, 3, len(dnames)), score=np.round(rng.uniform(0.05, 1.0, len(dnames)), 2))
        gts.append(gt)
        dts.append(dt)
    return gts, dts


def to_np(a):
    return {k: (np.array(v) if k == "name" else np.asarray(v)) for k, v in a.items()}


def jsonable(a):
    return {k: (list(v) if k == "name" else np.asarray(v).tolist()) for k, v in a.items()}


def main():
    _shim()
    from src.core import eval_utils as ref  # the reference's own module

    rng = np.random.default_rng(20)
    out = {"cases": []}
    for n_img in (50, 60, 113):   # fewer than num_parts = 50 images crash the reference (get_split_parts has no guard, eval_utils.py:332-339)
        gts, dts = make_annos(rng, n_img)
        g_np, d_np = [to_np(g) for g in gts], [to_np(d) for d in dts]
        text, m = ref.get_official_eval_result(g_np, d_np, [0, 1, 2], return_data=True)
        mo = np.array([[[0.7, 0.5, 0.5]] * 3, [[0.5, 0.25, 0.25]] * 3])
        ret = ref.eval_class(g_np, d_np, [0, 1, 2], [0, 1, 2], 0, mo)
        clean = [[ref.clean_data(g, d, c, diff) for c in (0, 1)] for g, d in zip(g_np[:5], d_np[:5]) for diff in (0, 2)]
        out["cases"].append(dict(gt=[jsonable(g) for g in gts], dt=[jsonable(d) for d in dts], text=text, map_bbox=np.asarray(m).tolist(),
                                 precision=np.nan_to_num(ret["precision"], nan=-1.0).tolist(),
                                 clean=[[[int(r[0]), [int(v) for v in r[1]], [int(v) for v in r[2]], len(r[3])] for r in row] for row in clean]))
    out["thresholds"] = []
    for n, ngt in ((5, 5), (30, 40), (120, 100), (3, 50)):
        s = np.round(rng.uniform(0, 1, n), 3)
        out["thresholds"].append(dict(scores=s.tolist(), num_gt=ngt, thresholds=[float(v) for v in ref.get_thresholds(s.copy(), ngt)]))
    # result-format geometry (pointpillars/src/core/box_ops.py: box_lidar_to_camera :521-546, boxes3d_kitti_camera_to_imageboxes
    # :653-668) on a KITTI-like calibration
    from src.core import box_ops

    rect = np.eye(4)
    rect[:3, :3] = [[0.9999, 0.0098, -0.0074], [-0.0099, 0.9999, -0.0043], [0.0074, 0.0044, 0.9999]]
    trv2c = np.eye(4)
    trv2c[:3, :] = [[0.0075, -0.9999, -0.0006, -0.0041], [0.0148, 0.0007, -0.9998, -0.0763], [0.9998, 0.0075, 0.0148, -0.2718]]
    p2 = np.eye(4)
    p2[:3, :] = [[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791], [0.0, 0.0, 1.0, 0.002745884]]
    n = 40
    boxes = np.stack([rng.uniform(2, 60, n), rng.uniform(-25, 25, n), rng.uniform(-2.2, -0.4, n), rng.uniform(0.5, 2.0, n),
                      rng.uniform(0.6, 4.5, n), rng.uniform(1.3, 1.9, n), rng.uniform(-3.2, 3.2, n)], 1)
    cam = box_ops.box_lidar_to_camera(boxes, rect, trv2c)
    img = box_ops.boxes3d_kitti_camera_to_imageboxes(cam, p2)
    out["geometry"] = dict(rect=rect.tolist(), trv2c=trv2c.tolist(), p2=p2.tolist(), boxes_lidar=boxes.tolist(), boxes_camera=np.asarray(cam).tolist(),
                           boxes_image=np.asarray(img).tolist())
    json.dump(out, open(os.path.join(HERE, "kitti_eval_vectors.json"), "w"))
    print("wrote", sum(len(c["gt"]) for c in out["cases"]), "images;", out["cases"][-1]["text"])


if __name__ == "__main__":
    main()
