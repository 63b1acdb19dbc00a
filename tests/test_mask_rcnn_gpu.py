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


def test_paste_masks_bit_exact_vs_oracle():
    """md_paste_masks (bit and uint8 forms, ragged widths) == oracle/np_ops.paste_masks bit for bit: boxes partly outside the image,
    tiny and huge boxes, empty slots, a degenerate box."""
    from minddet_amd import det_ops
    from oracle import np_ops

    rng = np.random.default_rng(5)
    for (H, W, B, D) in ((75, 101, 2, 12), (64, 96, 1, 7), (800, 1344, 1, 4)):
        S = 28
        masks = rng.uniform(0, 1, (B, D, S, S)).astype(np.float32)
        dets = np.zeros((B, D, 6), np.float32)
        cx, cy = rng.uniform(0, W, (B, D)), rng.uniform(0, H, (B, D))
        bw, bh = np.exp(rng.uniform(np.log(2), np.log(W), (B, D))), np.exp(rng.uniform(np.log(2), np.log(H), (B, D)))
        dets[..., 0], dets[..., 1], dets[..., 2], dets[..., 3] = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2
        dets[..., 4] = rng.uniform(0.05, 1, (B, D))
        dets[0, 1, 4] = 0.0
        dets[0, 2, 2] = dets[0, 2, 0]
        ref = np_ops.paste_masks(masks.reshape(-1, S, S), dets.reshape(-1, 6), (H, W), 0.5).reshape(B, D, H, W)
        md, dd = torch.from_numpy(masks).to(DEV), torch.from_numpy(dets).to(DEV)
        words = det_ops.paste_masks(md, dd, (H, W), 0.5, bits=True)
        assert words.shape == (B, D, H, (W + 31) // 32) and words.dtype == torch.int32
        np.testing.assert_array_equal(det_ops.unpack_mask_bits(words, W).cpu().numpy(), ref)
        u8 = det_ops.paste_masks(md, dd, (H, W), 0.5, bits=False)
        np.testing.assert_array_equal(u8.cpu().numpy(), ref)
        assert ref[0, 1].sum() == 0 and ref[0, 2].sum() == 0 and ref.sum() > 0
    from minddet_amd import _lib
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):      # output shape must match the attributes
        _lib.call("md_paste_masks", [md.view(-1, 28, 28), dd.view(-1, 6), torch.zeros((4, 800, 41), dtype=torch.int32, device=DEV)],
                  extra=det_ops._PasteAttrs(800, 1344, 0.5, 1))


def test_mask_rcnn_r101_fpn_shard_batch_8():
    """BASELINE.json configs[4] at the per-GPU shard bench.py runs (batch 8 in the stem layout), with the masks pasted to image resolution:
    structure, determinism, proposals / second stage of sampled images bit-exact from the device tensors (tests/stage_checks.py), prefix
    flags, masks zero past count, and the pasted bit masks of sampled detections == the oracle's, bit for bit."""
    from minddet.models import Config, build_detector
    from minddet_amd import det_ops, nn_ops
    from minddet_amd.data import synthetic_images
    from oracle import np_ops
    from tests import stage_checks

    cfg = Config.fromfile("configs/mask_rcnn/mask_rcnn_r101_fpn.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    B, H, W = 8, 800, 1344
    x = nn_ops.to_stem_layout(synthetic_images(B, H, W, seed=20240317, device=DEV))
    dets, count, masks, pasted, aux = m.forward(x, return_aux=True, paste=True)
    dets2, count2, masks2, pasted2 = m.forward(x, paste=True)
    torch.cuda.synchronize()
    assert dets.shape == (B, 100, 6) and masks.shape == (B, 100, 28, 28) and pasted.shape == (B, 100, H, W // 32)
    assert torch.equal(dets, dets2) and torch.equal(count, count2) and torch.equal(masks, masks2) and torch.equal(pasted, pasted2)
    d, c = stage_checks.structure(dets, count, 100, (H, W))
    assert c.sum() > 0
    stage_checks.rpn_images(m, aux, (0, 7))
    stage_checks.roi_images(m, aux, dets, count, (H, W), (0, 7))
    mk = masks.cpu().numpy()
    assert mk.min() >= 0.0 and mk.max() <= 1.0
    for b in range(B):
        assert (mk[b, c[b]:] == 0).all()      # (random-init R101 logits saturate: sigmoid gives exact 0 / 1 inside valid slots too)
        assert not bool((pasted[b, c[b]:] != 0).any())                       # empty slots paste nothing
    for b in (0, 7):
        n = min(int(c[b]), 5)
        if n == 0:
            continue
        ref = np_ops.paste_masks(mk[b, :n], d[b, :n], (H, W), m.mask_thr)
        got = det_ops.unpack_mask_bits(pasted[b, :n], W).cpu().numpy()
        np.testing.assert_array_equal(got, ref)
        for i in range(n):   # set pixels lie inside the box's pixel support
            ys, xs = np.nonzero(got[i])
            if len(ys):
                assert ys.min() >= np.floor(d[b, i, 1]) - 1 and ys.max() <= np.ceil(d[b, i, 3]) + 1
                assert xs.min() >= np.floor(d[b, i, 0]) - 1 and xs.max() <= np.ceil(d[b, i, 2]) + 1


def _damped_r101(cfg_path, gamma=0.2):
    """Mask R-CNN R101-FPN whose residual branches end in a BatchNorm with gamma = 0.2: with gamma = 1 the variance doubles in each of the 33
    random-init blocks and every mask logit saturates (sigmoid gives exact 0 / 1), so a full-size test could compare only the masks' support.
    Damped, the pyramid stays O(1) and the 28 x 28 mask VALUES are compared with the oracle."""
    from minddet.models import Config, build_detector

    cfg = Config.fromfile(cfg_path)
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg)
    for st in m.backbone.stages:
        for b in st:
            g_, beta, mean, var, eps = b.conv3.bn
            b.conv3.bn = (g_ * gamma, beta, mean, var, eps)
    return m.to(DEV), cfg


def test_mask_rcnn_r101_fpn_bench_batch_32_values_and_two_streams():
    """every quoted Mask R-CNN number is batch 32 -- so batch 32 is what runs here: one stream and the config's two
    streams (graphs.SplitForward) bit-identical incl. the pasted masks; proposals / second stage of sampled images bit-exact from the device
    tensors; and, on a non-saturating init, the mask-head VALUES of sampled detections against the torch-CPU oracle fed the device pyramid
    (tolerance as the tiny-config test: 6 bf16 conv layers + sigmoid, |dmask| <= 3e-2) and their pasted bit masks bit for bit."""
    from minddet_amd import det_ops, nn_ops
    from minddet_amd.data import synthetic_images
    from oracle import np_ops
    from tests import stage_checks

    m, cfg = _damped_r101("configs/mask_rcnn/mask_rcnn_r101_fpn.py")
    assert m.streams == 2
    B, H, W = 32, 800, 1344
    x = nn_ops.to_stem_layout(synthetic_images(B, H, W, seed=4242, device=DEV))
    dets, count, masks, pasted, aux = m.forward(x, return_aux=True, paste=True)
    torch.cuda.synchronize()
    assert dets.shape == (B, 100, 6) and masks.shape == (B, 100, 28, 28) and pasted.shape == (B, 100, H, W // 32)
    for rep in range(2):                                   # the two-stream path the bench line runs (16 + 16 images)
        d2, c2, m2, p2 = m.forward_split(x, paste=True)
        torch.cuda.synchronize()
        assert torch.equal(d2, dets) and torch.equal(c2, count) and torch.equal(m2, masks) and torch.equal(p2, pasted), rep
    d, c = stage_checks.structure(dets, count, 100, (H, W))
    assert c.sum() > 0
    stage_checks.rpn_images(m, aux, (0, 17, 31))
    stage_checks.roi_images(m, aux, dets, count, (H, W), (0, 17, 31))
    mk = masks.cpu().numpy()
    inner = []
    for b in (0, 17, 31):
        n = min(int(c[b]), 6)
        if n == 0:
            continue
        feats = [f[b:b + 1].float().cpu().permute(0, 3, 1, 2).contiguous() for f in aux["feats"]]
        dd = d[b:b + 1, :n].copy()
        ref = nets.mask_head_forward(m.mask_head, feats, dd, quant=True)[0]
        assert np.abs(mk[b, :n] - ref).max() <= 3e-2, (b, np.abs(mk[b, :n] - ref).max())
        inner.append(((mk[b, :n] > 1e-3) & (mk[b, :n] < 1 - 1e-3)).mean())
        got = det_ops.unpack_mask_bits(pasted[b, :n], W).cpu().numpy()
        np.testing.assert_array_equal(got, np_ops.paste_masks(mk[b, :n], d[b, :n], (H, W), m.mask_thr))
    assert inner and min(inner) > 0.5, inner              # the values compared are real probabilities, not saturated 0 / 1
    for b in range(B):
        assert (mk[b, c[b]:] == 0).all() and not bool((pasted[b, c[b]:] != 0).any())
