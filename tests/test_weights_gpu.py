"""-m gpu: a checkpoint exported from one CenterNet and imported into another (different random init) through the
MindSpore .ckpt codec gives bit-identical detections."""
import pytest
import torch

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


def test_imported_checkpoint_reproduces_the_detections(tmp_path):
    from minddet_amd import graphs, weights

    a = graphs.CenterNet(depth=18, num_classes=80, seed=11)
    b = graphs.CenterNet(depth=18, num_classes=80, seed=12)
    p = str(tmp_path / "a.ckpt")
    weights.write_ms_ckpt(p, weights.centernet_state(a, "ms"))
    assert weights.load_centernet(b, weights.read_ms_ckpt(p)) == []
    a.to(DEV)
    b.to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.zeros((1, 128, 192, 8))
    x[..., :3] = torch.randn((1, 128, 192, 3), generator=g)
    xb = x.to(torch.bfloat16).to(DEV)
    da, db = a.forward(xb), b.forward(xb)
    for ta, tb in zip(da, db):
        assert torch.equal(ta, tb)
