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
