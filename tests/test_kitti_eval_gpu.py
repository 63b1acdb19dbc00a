"""-m gpu: the KITTI evaluator with its product overlap source (md_rotate_iou_eval on the GPU, where the reference calls its
numba-CUDA kernel) gives the same precision rows as with the CPU oracle's rotated overlaps injected."""
import numpy as np
import pytest

import oracle
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]


def _annos(rng, n_img, with_score):
    out = []
    for _ in range(n_img):
        n = int(rng.integers(1, 6))
        loc = np.stack([rng.uniform(-15, 15, n), np.full(n, 1.6), rng.uniform(5, 45, n)], 1)
        dims = np.stack([rng.uniform(3.2, 4.6, n), rng.uniform(1.4, 1.7, n), rng.uniform(1.5, 1.8, n)], 1)
        u = 600 + 700 * loc[:, 0] / loc[:, 2]
        half = 700 * 2.0 / loc[:, 2]
        bbox = np.stack([u - half, 180 - 40 - 600 / loc[:, 2], u + half, 180 + 600 / loc[:, 2]], 1)
        a = dict(name=np.array(["Car"] * n), bbox=bbox, location=loc, dimensions=dims, rotation_y=rng.uniform(-3, 3, n),
                 occluded=rng.integers(0, 3, n), truncated=rng.choice([0.0, 0.2, 0.4], n), alpha=rng.uniform(-3, 3, n))
        if with_score:
            a["score"] = rng.uniform(0.1, 1.0, n)
        out.append(a)
    return out


def test_device_overlaps_give_the_oracle_precision_rows():
    from minddet_amd import kitti_eval as ke

    rng = np.random.default_rng(5)
    gt = _annos(rng, 24, False)
    dt = []
    for g in gt:                       # detections = jittered ground truth + one stray box
        d = {k: np.array(v) for k, v in g.items()}
        d["location"] = d["location"] + rng.normal(0, 0.15, d["location"].shape)
        d["rotation_y"] = d["rotation_y"] + rng.normal(0, 0.05, d["rotation_y"].shape)
        d["score"] = rng.uniform(0.1, 1.0, len(d["name"]))
        dt.append(d)
    rot = lambda b, q, c: oracle.rotate_iou_eval(np.ascontiguousarray(b, np.float32), np.ascontiguousarray(q, np.float32), c)
    mo = np.full((1, 3, 1), 0.5)
    for metric in (1, 2):
        dev = ke.eval_class(gt, dt, [0], [0, 1, 2], metric, mo)
        ref = ke.eval_class(gt, dt, [0], [0, 1, 2], metric, mo, rotate_iou=rot)
        assert np.allclose(dev["precision"], ref["precision"], atol=1e-6, equal_nan=True)
        assert np.allclose(dev["recall"], ref["recall"], atol=1e-6, equal_nan=True)
        assert dev["precision"].max() > 0.5
    text, out = ke.get_official_eval_result(gt, dt, ["Car"])
    assert "Car_3d/moderate_R40" in out and "bev AP:" in text
