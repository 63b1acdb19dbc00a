"""-m gpu: Mask R-CNN (BASELINE configs[4]) -- the mask branch against the torch-CPU oracle on the tiny config, fed the
DEVICE pyramid and detections at the stage boundary (so only the mask head's own arithmetic is compared), plus structural
properties at the R101-FPN size.  Tolerance: 6 bf16 conv layers + sigmoid: |dmask| <= 3e-2.  Parity unpinned (absent
from the reference)."""
import numpy as np
import pytest
import torch

from oracle import nets
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


def test_mask_branch_vs_oracle_tiny():
    from minddet.models import Config, build_detector

    cfg = Config.fromfile("configs/mask_rcnn/mask_rcnn_tiny.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.zeros((2, 128, 192, 8))
    x[..., :3] = torch.randn((2, 128, 192, 3), generator=g)
    dets, count, masks, aux = m.forward(x.to(torch.bfloat16).to(DEV), return_aux=True)
    torch.cuda.synchronize()
    assert masks.shape == (2, 20, 28, 28) and int(count.sum()) > 0
    feats = [f.float().cpu().permute(0, 3, 1, 2).contiguous() for f in aux["feats"]]
    ref = nets.mask_head_forward(m.mask_head, feats, dets.cpu().numpy(), quant=True)
    got = masks.cpu().numpy()
    assert np.abs(got - ref).max() <= 3e-2, np.abs(got - ref).max()
    c = count.cpu().numpy()
    for b in range(2):
        assert (got[b, c[b]:] == 0).all() and (got[b, :c[b]] > 0).all()   # empty slots: zero masks


def test_mask_rcnn_r101_fpn_full_size_structure():
    from minddet.models import Config, build_detector
    from minddet_amd import nn_ops
    from minddet_amd.data import synthetic_images

    cfg = Config.fromfile("configs/mask_rcnn/mask_rcnn_r101_fpn.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    x = nn_ops.to_stem_layout(synthetic_images(2, 800, 1344, device=DEV))
    dets, count, masks = m.forward(x)
    dets2, count2, masks2 = m.forward(x)
    torch.cuda.synchronize()
    assert dets.shape == (2, 100, 6) and masks.shape == (2, 100, 28, 28)
    assert torch.equal(dets, dets2) and torch.equal(masks, masks2)          # deterministic
    assert float(masks.min()) >= 0.0 and float(masks.max()) <= 1.0
