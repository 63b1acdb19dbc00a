"""-m gpu: md_assign_targets against the reference's own create_target_np outputs (tests/golden/target_vectors.npz) and,
at the PointPillars car size (107 136 anchors), against the oracle restatement that those vectors pin.
labels / weights / gt ids bit-exact; the three log() size targets within 4 ulp (device logf vs numpy's), the rest exact."""
import os

import numpy as np
import pytest
import torch

from oracle import np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"
G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "target_vectors.npz"))


def _check_targets(got, ref):
    np.testing.assert_array_equal(got[:, [0, 1, 2, 6]], ref[:, [0, 1, 2, 6]])
    a, b = got[:, 3:6], ref[:, 3:6]
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert (ulp <= 4).all() or np.allclose(a, b, rtol=0, atol=1e-7), ulp.max()


def _run(anchors, gt, cls, mt, ut, mask):
    from minddet_amd import det_ops

    T = lambda x, dt=None: torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
    out = det_ops.assign_targets(T(anchors), T(gt), T(cls), T(mt), T(ut), None if mask is None else T(mask))
    return [o.cpu().numpy() for o in out]


@pytest.mark.parametrize("case", ["car", "nomask", "nogt", "pedcyc"])
def test_device_equals_reference_vectors(case):
    mask = G[case + "_mask"]
    labels, targets, weights, gt_ids = _run(G[case + "_anchors"], G[case + "_gt"], G[case + "_cls"], G[case + "_mt"], G[case + "_ut"],
                                            mask if mask.size else None)
    np.testing.assert_array_equal(labels, G[case + "_labels"])
    np.testing.assert_array_equal(weights, G[case + "_weights"])
    _check_targets(targets, G[case + "_targets"])
    assert ((gt_ids >= 0) == (labels > 0)).all()


def test_pointpillars_car_size_vs_oracle():
    anchors = np_ops.create_anchors_3d_stride((1, 248, 216), sizes=(1.6, 3.9, 1.56), anchor_strides=(0.32, 0.32, 0.0),
                                              anchor_offsets=(0.16, -39.52, -1.78), rotations=(0, 1.57)).reshape(-1, 7).astype(np.float32)
    assert anchors.shape[0] == 107136
    rng = np.random.default_rng(5)
    gt = np.zeros((23, 7), np.float32)
    gt[:, 0] = rng.uniform(0, 69.12, 23); gt[:, 1] = rng.uniform(-39.68, 39.68, 23); gt[:, 2] = rng.uniform(-2, 0, 23)
    gt[:, 3] = rng.uniform(1.4, 1.9, 23); gt[:, 4] = rng.uniform(3.2, 4.6, 23); gt[:, 5] = rng.uniform(1.3, 1.8, 23)
    gt[:, 6] = rng.uniform(-np.pi, np.pi, 23)
    mask = (rng.uniform(0, 1, anchors.shape[0]) < 0.2).astype(np.uint8)
    mt = np.full((anchors.shape[0],), 0.6, np.float32)
    ut = np.full((anchors.shape[0],), 0.45, np.float32)
    cls = np.ones((23,), np.int32)
    labels, targets, weights, gt_ids = _run(anchors, gt, cls, mt, ut, mask)
    rl, rt, rw, rg = np_ops.create_target(anchors, gt, cls, mt, ut, mask.astype(bool))
    np.testing.assert_array_equal(labels, rl)
    np.testing.assert_array_equal(weights, rw)
    np.testing.assert_array_equal(gt_ids, rg)
    _check_targets(targets, rt)
    assert (labels > 0).sum() > 20
