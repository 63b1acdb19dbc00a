"""-m gpu: a captured HIP-graph step (minddet_amd/replay.py) reproduces the eager step bit for bit, also on a batch it was not
captured with; shape mismatches are rejected."""
import pytest
import torch

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


def _batch(seed, shape):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(shape)
    x[..., :3] = torch.randn(shape[:3] + (3,), generator=g)
    return x.to(torch.bfloat16).to(DEV)


@pytest.mark.parametrize("which", ["faster_rcnn_tiny", "yolov8_tiny", "centernet"])
def test_captured_step_equals_eager(which):
    from minddet.models import Config, build_detector
    from minddet_amd import graphs
    from minddet_amd.replay import CapturedStep

    if which == "centernet":
        m, shape = graphs.CenterNet(depth=18, num_classes=80, seed=3).to(DEV), (2, 128, 192, 8)
    else:
        cfg = Config.fromfile({"faster_rcnn_tiny": "configs/faster_rcnn/faster_rcnn_tiny.py", "yolov8_tiny": "configs/yolov8/yolov8_tiny.py"}[which])
        m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
        shape = (2,) + tuple(cfg.data.input_hw) + (8,)

    def fwd(x):
        out = m.forward(x)
        return out if isinstance(out, (tuple, list)) else (out,)

    a, b = _batch(1, shape), _batch(2, shape)
    step = CapturedStep(fwd, a)
    for x in (b, a, b):
        eager = [t.clone() for t in fwd(x)]
        torch.cuda.synchronize()
        got = step(x)
        torch.cuda.synchronize()
        assert len(got) == len(eager)
        for tg, te in zip(got, eager):
            assert torch.equal(tg, te)
    with pytest.raises(ValueError):
        step(_batch(3, (1,) + shape[1:]))


def test_rpn_side_stream_overlap_is_bit_identical():
    """graphs.RPN_OVERLAP (per-level proposal selection on a side stream behind the level's conv): detections equal the single-stream
    path bit for bit, eager and under CapturedStep, on several batches in a row (a stream hazard would show as a mismatch)."""
    from minddet.models import Config, build_detector
    from minddet_amd import graphs
    from minddet_amd.replay import CapturedStep

    cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_tiny.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    shape = (3,) + tuple(cfg.data.input_hw) + (8,)
    batches = [_batch(s, shape) for s in (11, 12, 13)]
    old = graphs.RPN_OVERLAP
    try:
        graphs.RPN_OVERLAP = False
        ref = []
        for x in batches:
            d, c = m.forward(x)
            ref.append((d.clone(), c.clone()))
        torch.cuda.synchronize()
        graphs.RPN_OVERLAP = True
        for _ in range(2):
            for x, (d0, c0) in zip(batches, ref):
                d, c = m.forward(x)
                assert torch.equal(d, d0) and torch.equal(c, c0)
        step = CapturedStep(lambda x: tuple(m.forward(x)), batches[0])
        for x, (d0, c0) in zip(batches, ref):
            d, c = step(x)
            torch.cuda.synchronize()
            assert torch.equal(d, d0) and torch.equal(c, c0)
    finally:
        graphs.RPN_OVERLAP = old
