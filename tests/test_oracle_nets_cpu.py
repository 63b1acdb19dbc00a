"""CPU: the torch/numpy oracle of the two-stage graph runs end to end on the tiny config, the
vectorised RoIAlign equals the loop form, and the model-build surface resolves by `type` string."""
import numpy as np
import torch

from minddet.models import Config, build_detector, BACKBONES, build_from_cfg
from oracle import nets, np_ops


def test_build_surface_and_macs():
    cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
    assert cfg.model.type == "FasterRCNN" and cfg.model.backbone.depth == 50
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    gmac = m.macs_per_image(800, 1344) / 1e9
    assert 210 < gmac < 222  # ~216 GMAC
    r18 = build_from_cfg(dict(type="ResNet", depth=18), BACKBONES)
    assert r18.out_channels == [64, 128, 256, 512]
    try:
        build_from_cfg(dict(type="NoSuchNet"), BACKBONES)
        assert False
    except KeyError:
        pass


def test_roi_align_fast_equals_loop():
    rng = np.random.default_rng(0)
    feat = rng.normal(0, 1, (6, 20, 30)).astype(np.float32)
    rois = np.array([[2, 3, 50, 40], [-10, -10, 20, 20], [100, 60, 130, 90], [0, 0, 119, 79], [5, 5, 5.5, 5.5]], np.float32)
    a = np_ops.roi_align(feat, rois, 7, 0.25, 2, True)
    b = np_ops.roi_align_fast(feat, rois, 7, 0.25, 2, True, chunk=2)
    np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-6)
    # known answer: constant feature map -> constant output; RoI fully outside -> 0
    c = np_ops.roi_align_fast(np.full((2, 8, 8), 3.0, np.float32), np.array([[4, 4, 20, 20]], np.float32), 2, 0.25, 2, True)
    np.testing.assert_allclose(c, 3.0, rtol=1e-6)
    z = np_ops.roi_align_fast(np.ones((1, 8, 8), np.float32), np.array([[400, 400, 420, 420]], np.float32), 2, 0.25, 2, True)
    assert (z == 0).all()


def test_tiny_faster_rcnn_oracle_runs():
    cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_tiny.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 128, 192, 8), generator=g)
    dets, count = nets.faster_rcnn_forward(m, x, quant=True)
    assert dets.shape == (2, 20, 6) and count.shape == (2,)
    for b in range(2):
        n = count[b]
        assert (np.diff(dets[b, :n, 4]) <= 0).all()          # score order
        assert (dets[b, n:] == 0).all()
        assert ((dets[b, :n, 5] >= 0) & (dets[b, :n, 5] < 5)).all()
        assert (dets[b, :n, 0] <= dets[b, :n, 2]).all() and (dets[b, :n, 2] <= 192).all()
