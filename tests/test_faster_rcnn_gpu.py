"""-m gpu: the two-stage path stage by stage against the CPU oracle on the tiny config (sizes the
oracle finishes in seconds), plus structural properties at the BASELINE size.

Conv stacks: fp tolerance (bf16 storage on both sides; rtol/atol 3e-2 of rms after ~10 layers).
Everything index-like downstream (top-k indices, NMS keep masks, RoI order, packed labels) is checked
BIT-EXACT by feeding the oracle the device tensors at the stage boundary (heads, boxes, scores, cand),
so a fp difference upstream cannot hide or fake an index mismatch."""
import numpy as np
import pytest
import torch

from oracle import nets, np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


@pytest.fixture(scope="module")
def tiny():
    from minddet.models import Config, build_detector

    cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_tiny.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.zeros((2, 128, 192, 8))
    x[..., :3] = torch.randn((2, 128, 192, 3), generator=g)
    xb = x.to(torch.bfloat16)
    dets, count, aux = m.forward(xb.to(DEV), return_aux=True)
    torch.cuda.synchronize()
    return m, xb, dets, count, aux


def test_backbone_fpn_features_vs_torch_fp32(tiny):
    m, xb, _, _, aux = tiny
    x = xb[..., :3].float().permute(0, 3, 1, 2).contiguous()
    ref = nets.fpn_forward(m.neck, nets.resnet_forward(m.backbone, x, quant=True), quant=True)
    for f_dev, f_ref in zip(aux["feats"], ref):
        got = f_dev.float().cpu().permute(0, 3, 1, 2)
        assert got.shape == f_ref.shape
        rms = f_ref.pow(2).mean().sqrt().item()
        err = (got - f_ref).abs().max().item()
        assert err <= 3e-2 * max(rms, 1e-3) + 3e-2 * f_ref.abs().max().item() * 0.1, (err, rms)


def test_stem_layout_path_matches_oracle():
    """A 64-wide ResNet-18-style backbone fed the batch in the stem layout (one md_stem_pool launch instead of
    conv + maxpool) against the torch-CPU oracle of the same graph, and against the two-launch path."""
    from minddet_amd import graphs, nn_ops

    bb = graphs.ResNet(depth=18, base_width=64, layers=[1, 1, 1, 1], seed=5).to(DEV)
    g = torch.Generator().manual_seed(1)
    x8 = torch.zeros((2, 64, 128, 8))
    x8[..., :3] = torch.randn((2, 64, 128, 3), generator=g)
    xb = x8.to(torch.bfloat16)
    x4 = nn_ops.to_stem_layout(xb.to(DEV))
    assert x4.shape == (2, 64 + 16, 128 + 16, 4)
    ref = nets.resnet_forward(bb, xb[..., :3].float().permute(0, 3, 1, 2).contiguous(), quant=True)
    two = bb(xb.to(DEV))
    for f_dev, f_two, f_ref in zip(bb(x4), two, ref):
        got = f_dev.float().cpu().permute(0, 3, 1, 2)
        rms = f_ref.pow(2).mean().sqrt().item()
        err = (got - f_ref).abs().max().item()
        assert err <= 3e-2 * max(rms, 1e-3) + 3e-2 * f_ref.abs().max().item() * 0.1, (err, rms)
        assert (f_dev.float() - f_two.float()).abs().max().item() <= 3e-2 * max(rms, 1e-3) + 3e-2 * f_ref.abs().max().item() * 0.1


def test_rpn_stage_indices_exact(tiny):
    m, xb, _, _, aux = tiny
    rpn = m.rpn_head
    H, W = xb.shape[1], xb.shape[2]
    heads = [h.float().cpu().numpy() for h in aux["rpn"]["heads"]]
    sizes = [(h.shape[1], h.shape[2]) for h in heads]
    anchors = np_ops.fpn_anchors(sizes, rpn.strides, rpn.scale, rpn.ratios)
    boxes_d = aux["rpn"]["boxes"].cpu().numpy()
    scores_d = aux["rpn"]["scores"].cpu().numpy()
    counts_d = aux["rpn"]["counts"].cpu().numpy()
    L, B, k = scores_d.shape
    o = 0
    for l, hd in enumerate(heads):
        n = sizes[l][0] * sizes[l][1] * rpn.A
        for b in range(B):
            idx, c = nets.rpn_level_select(hd[b], rpn.A, k)
            assert counts_d[l, b] == c == min(k, n)
            bx, sc = nets.rpn_level_decode(hd[b], anchors[o:o + n], idx, rpn.A, (H, W))
            np.testing.assert_allclose(boxes_d[l, b, :c], bx, rtol=1e-5, atol=2e-3)   # => same anchors selected
            np.testing.assert_allclose(scores_d[l, b, :c], sc, rtol=1e-5, atol=1e-6)
            assert (boxes_d[l, b, c:] == 0).all()
        o += n
    # NMS + merge + per-image top-k from the DEVICE boxes/scores: exact
    keep_o, topi_o, rois_o, cnt_o = nets.proposals_from_lists(boxes_d, scores_d, counts_d, rpn.nms_thr, rpn.max_per_img)
    np.testing.assert_array_equal(aux["rpn"]["keep"].cpu().numpy().reshape(L, B, k), keep_o)
    np.testing.assert_array_equal(aux["roi_cnt"].cpu().numpy(), cnt_o)
    np.testing.assert_array_equal(aux["rois"].cpu().numpy(), rois_o)


def test_roi_stage(tiny):
    m, xb, dets, count, aux = tiny
    roi, post = m.roi_head, m.rpn_head.max_per_img
    H, W = xb.shape[1], xb.shape[2]
    rois = aux["rois"].cpu().numpy()
    cnt = aux["roi_cnt"].cpu().numpy()
    feats = [f.float().cpu().numpy().transpose(0, 3, 1, 2) for f in aux["feats"][:4]]
    pooled = aux["roi"]["pooled"].float().cpu().numpy()
    lv = np.clip(np_ops.fpn_level(rois[:, 1:]) - 2, 0, 3)
    checked = 0
    for r in range(0, rois.shape[0], 7):
        if (r % post) >= cnt[r // post]:
            continue
        ref = np_ops.roi_align(feats[lv[r]][int(rois[r, 0])], rois[r:r + 1, 1:], roi.P, 1.0 / roi.strides[lv[r]], 2, True)[0]
        # the FPN level is an exact integer map on both sides, so EVERY sampled RoI pools from the oracle's level
        err = np.abs(pooled[r] - ref.transpose(1, 2, 0)).max()
        assert err <= 1e-2 * (1 + np.abs(ref).max()), (r, int(lv[r]), float(err))
        checked += 1
    assert checked > 0
    # FC stack on the device's pooled features vs fp32 torch (bf16-rounded weights)
    x2 = torch.from_numpy(pooled.reshape(rois.shape[0], -1))
    for mod in (roi.fc1, roi.fc2, roi.fc_out):
        w, b = nets.fold(mod)
        x2 = x2 @ w.to(torch.bfloat16).float().view(w.shape[0], -1).t() + b
        if mod.relu:
            x2 = torch.relu(x2)
        x2 = x2.to(torch.bfloat16).float()
    cls_reg_d = aux["roi"]["cls_reg"].float().cpu()
    n_out = roi.reg_offset + 4 * roi.nc
    err = (cls_reg_d[:, :n_out] - x2[:, :n_out]).abs().max().item()
    assert err <= 3e-2 * (1 + x2.abs().max().item()), err
    # candidate scores from the DEVICE logits (softmax fp tolerance), then everything after from the DEVICE cand: exact
    cr = cls_reg_d.numpy()
    cand_o = nets.rcnn_candidates(cr, cnt, roi.nc, roi.score_thr, post)
    cand_d = aux["roi"]["cand"].cpu().numpy()
    cand_d = np.where(cand_d < -1e30, -np.inf, cand_d)
    both = np.isfinite(cand_o) & np.isfinite(cand_d)
    assert (np.isfinite(cand_o) != np.isfinite(cand_d)).mean() < 1e-3  # threshold flips at 1 ulp only
    np.testing.assert_allclose(cand_d[both], cand_o[both], rtol=2e-6, atol=1e-7)
    dets_o, count_o, sel = nets.rcnn_finish(cand_d, cr, rois, roi.nc, roi.reg_offset, (H, W), roi.nms_pre, roi.nms_thr,
                                            roi.max_per_img, post)
    np.testing.assert_array_equal(count.cpu().numpy(), count_o)
    d = dets.cpu().numpy()
    np.testing.assert_array_equal(d[..., 5], dets_o[..., 5])                  # labels exact
    np.testing.assert_array_equal(d[..., 4], dets_o[..., 4])                  # scores are the device cand values
    np.testing.assert_allclose(d[..., :4], dets_o[..., :4], rtol=1e-5, atol=2e-3)
    for b in range(len(sel)):
        m_ = len(sel[b]["idx"])
        np.testing.assert_array_equal(aux["roi"]["sel_idx"].cpu().numpy()[b, :m_], sel[b]["idx"])
        k_o = sel[b]["keep"] & (np.cumsum(sel[b]["keep"]) <= roi.max_per_img)  # the device scan stops at max_output
        np.testing.assert_array_equal(aux["roi"]["keep"].cpu().numpy()[b, :m_].astype(bool), k_o)
    assert count_o.sum() > 0  # the test actually exercises the class-wise NMS


def test_full_size_structural_properties():
    """BASELINE size (R50-FPN, 800x1344, batch 2): shapes, padding, ordering, determinism."""
    from minddet.models import Config, build_detector
    from minddet_amd.data import synthetic_images

    cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    x = synthetic_images(2, 800, 1344, device=DEV)
    dets, count, aux = m.forward(x, return_aux=True)
    dets2, count2 = m.forward(x)
    torch.cuda.synchronize()
    assert [tuple(f.shape[1:3]) for f in aux["feats"]] == [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    assert dets.shape == (2, 100, 6) and aux["rois"].shape == (2000, 5)
    assert torch.equal(dets, dets2) and torch.equal(count, count2)              # deterministic
    # the same batch in the stem layout (fused stem kernel): same shapes, (nearly) the same detections
    from minddet_amd import nn_ops
    dets4, count4 = m.forward(nn_ops.to_stem_layout(x))
    # (md_stem_pool accumulates the 7x7 window in another K order than md_conv2d + md_maxpool2d: last-bit differences in the stem's bf16
    # output; measured on this batch (r03): identical counts, top-score difference 0.0)
    cdiff = (count4.cpu() - count.cpu()).abs().max().item()
    sdiff = (dets4[:, 0, 4] - dets[:, 0, 4]).abs().max().item()
    print("stem-layout vs 8-channel path: max count difference", cdiff, "max top-score difference", sdiff)
    assert dets4.shape == dets.shape and cdiff == 0 and sdiff <= 1e-3
    d, c = dets.cpu().numpy(), count.cpu().numpy()
    for b in range(2):
        n = c[b]
        assert (np.diff(d[b, :n, 4]) <= 0).all() and (d[b, n:] == 0).all()
        assert (d[b, :n, 0] >= 0).all() and (d[b, :n, 2] <= 1344).all() and (d[b, :n, 3] <= 800).all()
    # idempotence of the per-level NMS at full size: survivors re-run through NMS all survive
    from minddet_amd import det_ops
    boxes, keep = aux["rpn"]["boxes"][0, 0], aux["rpn"]["keep"].view(5, 2, -1)[0, 0].bool()
    kept = boxes[keep].contiguous()
    mask2, _, num2 = det_ops.nms_aligned(kept, 0.7, mode=2)
    assert int(num2[0]) == kept.shape[0] and bool(mask2.all())


def test_end_to_end_agreement_in_map_units():
    """The whole device path against the whole CPU oracle (independent runs, nothing shared at stage boundaries): the oracle's
    detections are the ground truth of a COCO bbox evaluation of the device detections (tests/agreement_ap.py).  Random-init
    weights make every score margin tiny, so one bf16 rounding can swap ranks or NMS survivors and a single 4-image sample swings
    between 0.87 and 1.0; the MEAN over six seeds of 8 images each is stable.  Measured r02: 0.970 against the oracle that rounds to
    bf16 exactly where the device stores bf16 (per seed 0.93-1.0), 0.911 against the pure fp32 oracle (0.81-0.95)."""
    from tests import agreement_ap

    rs = [agreement_ap.agreement(B=8, seed=seed) for seed in range(6)]
    per_seed = [(round(r["bf16-matched oracle"]["AP"], 3), round(r["fp32 oracle"]["AP"], 3)) for r in rs]
    print("AP per seed (bf16-matched oracle, fp32 oracle):", per_seed)
    assert all(r["bf16-matched oracle"]["n_oracle"] > 40 for r in rs)
    ap_q = sum(r["bf16-matched oracle"]["AP"] for r in rs) / len(rs)
    ar_q = sum(r["bf16-matched oracle"]["AR100"] for r in rs) / len(rs)
    ap_f = sum(r["fp32 oracle"]["AP"] for r in rs) / len(rs)
    assert ap_q >= 0.93 and ar_q >= 0.94, per_seed
    assert min(r["bf16-matched oracle"]["AP"] for r in rs) >= 0.85, per_seed
    # the independent fp32 oracle: measured 0.911 over these six seeds (r02 / r03); the floor is that value minus the seed-to-seed noise of the mean
    assert ap_f >= 0.89, per_seed


def test_deep_bottleneck_stack_matches_oracle():
    """A 64-wide Bottleneck ResNet, ~25 bf16-rounded layers deep, against the torch-CPU oracle."""
    from minddet_amd import graphs

    bb = graphs.ResNet(depth=50, base_width=64, layers=[3, 2, 1, 1], seed=9).to(DEV)
    g = torch.Generator().manual_seed(4)
    x8 = torch.zeros((2, 96, 160, 8))
    x8[..., :3] = torch.randn((2, 96, 160, 3), generator=g)
    xb = x8.to(torch.bfloat16)
    feats = bb(xb.to(DEV))
    ref = nets.resnet_forward(bb, xb[..., :3].float().permute(0, 3, 1, 2).contiguous(), quant=True)
    for f_c, f_ref in zip(feats, ref):
        rms = f_ref.pow(2).mean().sqrt().item()
        # the distance to the oracle is judged as an rms (2 %) with a loose cap on single values
        d = f_c.float().cpu().permute(0, 3, 1, 2) - f_ref
        assert d.pow(2).mean().sqrt().item() <= 2e-2 * max(rms, 1e-3)
        assert d.abs().max().item() <= 0.1 * f_ref.abs().max().item()


def _second_stage_device(head, cls_reg, rois, roi_cnt, img_hw):
    from minddet_amd import det_ops

    head.prefix_status.clear()
    B = roi_cnt.numel()
    post = rois.shape[0] // B
    cand = det_ops.rcnn_scores(cls_reg, roi_cnt, head.nc, head.score_thr)
    seg = torch.arange(0, (B + 1) * post * head.nc, post * head.nc, dtype=torch.int32, device=DEV)
    sv, si, sc = det_ops.topk_segmented(cand, seg, head.nms_pre, max_segment=post * head.nc)
    boxes, labels = det_ops.rcnn_decode_selected(cls_reg, rois, si, sc, head.nc, head.reg_offset, img_hw)
    keep, kidx, num = det_ops.nms_aligned(boxes, head.nms_thr, mode=det_ops.NMS_MODE_STRICT, count=sc, group=labels, max_output=head.max_per_img)
    dets, count = det_ops.pack_detections(boxes, sv, labels, kidx, num, head.max_per_img, sel_cnt=sc, status=head.prefix_status.tensor(B, DEV))
    return dets.cpu().numpy(), count.cpu().numpy(), cand.cpu().numpy(), head.prefix_status.tensor(B, DEV).cpu().numpy()


def test_pre_nms_prefix_against_the_untruncated_oracle():
    """The class-wise NMS of the second stage runs on the top-nms_pre candidates.  With MORE candidates above the threshold than
    nms_pre: (a) when the prefix yields max_per_img survivors the result equals the UNTRUNCATED definition (every candidate above
    score_thr -> class-wise NMS -> top max_per_img) and no flag is raised; (b) when the full prefix runs out before the quota the
    result can differ -- and exactly then the sticky device flag of that image is set; (c) a prefix that was not full never flags."""
    from minddet_amd import graphs

    nc, post, B = 4, 96, 3
    head = graphs.StandardRoIHead(in_channels=8, fc_channels=8, num_classes=nc, roi_size=2, score_thr=0.05, nms_thr=0.5, max_per_img=10,
                                  nms_pre=64, featmap_strides=(4,))
    rng = np.random.default_rng(21)
    R = B * post
    Cp = head.reg_offset + 4 * nc
    cls_reg = np.zeros((R, (Cp + 7) // 8 * 8), np.float32)
    rois = np.zeros((R, 5), np.float32)
    rois[:, 0] = np.repeat(np.arange(B), post)
    # image 0: well separated RoIs (a 12 x 8 grid of 20-px boxes 40 px apart): nothing suppresses anything -> the quota is
    #          reached inside the prefix although 96 * 2 = 192 candidates exceed nms_pre = 64
    gx, gy = np.meshgrid(np.arange(12) * 40.0, np.arange(8) * 40.0)
    rois[:post, 1], rois[:post, 2] = gx.ravel(), gy.ravel()
    rois[:post, 3], rois[:post, 4] = gx.ravel() + 20, gy.ravel() + 20
    # image 1: the 70 best-scored RoIs are near-copies of ONE box (class 0): the full prefix collapses to one survivor per class,
    #          the separated RoIs behind it would fill the quota -> the cut changes the result, the flag must be raised
    rois[post:2 * post, 1:] = rois[:post, 1:]
    rois[post:post + 70, 1:] = np.array([100, 100, 160, 160], np.float32) + rng.uniform(-1, 1, (70, 4)).astype(np.float32)
    # image 2: only 20 valid RoIs (40 candidates < nms_pre): never flagged
    rois[2 * post:, 1:] = rois[:post, 1:]
    roi_cnt = np.array([post, post, 20], np.int32)
    # two classes above the threshold per RoI (distinct scores), background last
    logits = np.full((R, nc + 1), -4.0, np.float32)
    logits[:, 0] = 2.0 + rng.permutation(R).astype(np.float32) * 1e-3
    logits[:, 1] = 1.0 + rng.permutation(R).astype(np.float32) * 1e-3
    logits[post:post + 70, 0] += 1.0   # image 1: its first 70 RoIs get the highest class-0 scores
    cls_reg[:, :nc + 1] = logits
    cr = torch.from_numpy(cls_reg).to(torch.bfloat16)
    img_hw = (400, 600)
    d, c, cand, status = _second_stage_device(head, cr.to(DEV), torch.from_numpy(rois).to(DEV), torch.from_numpy(roi_cnt).to(DEV), img_hw)
    crf = cr.float().numpy()
    cand = np.where(cand < -1e30, -np.inf, cand)
    assert (np.isfinite(cand).sum(1) > head.nms_pre).tolist() == [True, True, False]
    full, full_c, _ = nets.rcnn_finish(cand, crf, rois, nc, head.reg_offset, img_hw, post * nc, head.nms_thr, head.max_per_img, post)   # untruncated
    cut, cut_c, _ = nets.rcnn_finish(cand, crf, rois, nc, head.reg_offset, img_hw, head.nms_pre, head.nms_thr, head.max_per_img, post)
    np.testing.assert_array_equal(c, cut_c)
    np.testing.assert_array_equal(d[..., 4:], cut[..., 4:])
    np.testing.assert_allclose(d[..., :4], cut[..., :4], rtol=1e-5, atol=2e-3)
    assert status.tolist() == [0, 1, 0]
    for b in (0, 2):   # flag clear => identical to the untruncated definition
        assert c[b] == full_c[b]
        np.testing.assert_array_equal(d[b, :, 4:], full[b, :, 4:])
    assert full_c[1] == head.max_per_img and c[1] < full_c[1]    # the flagged image is the one that differs
    # the flag is sticky until cleared, and a clean batch does not set it
    assert head.prefix_status.flagged() == 1
    head.prefix_status.clear()
    assert head.prefix_status.flagged() == 0


def test_benchmark_batch_60_two_stage_forward():
    """The benchmark's real shape: R50-FPN 800x1344 at batch 60 in the stem layout -- the P2-level tensors are 2.06 GB, so md_conv2d's
    > 2 GiB image chunking and every launch grid of bench.py's step run here; structure, determinism, chunk-independence."""
    from minddet.models import Config, build_detector
    from minddet_amd import nn_ops
    from minddet_amd.data import synthetic_images

    cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    x = nn_ops.to_stem_layout(synthetic_images(60, 800, 1344, seed=20240317, device=DEV))
    dets, count = m.forward(x)
    dets2, count2 = m.forward(x)
    torch.cuda.synchronize()
    assert dets.shape == (60, 100, 6) and count.shape == (60,)
    assert torch.equal(dets, dets2) and torch.equal(count, count2)
    d, c = dets.cpu().numpy(), count.cpu().numpy()
    for b in range(60):
        n = c[b]
        assert 0 <= n <= 100 and (np.diff(d[b, :n, 4]) <= 0).all() and (d[b, n:] == 0).all()
        assert (d[b, :n, 0] >= 0).all() and (d[b, :n, 2] <= 1344).all() and (d[b, :n, 3] <= 800).all()
    # images are independent: the first four images alone give (nearly: the dispatcher may pick other kernels for the smaller
    # grids, bf16 rounding then differs in the last bit) the same detections as inside the batch of 60
    d4, c4 = m.forward(x[:4].contiguous())
    torch.cuda.synchronize()
    assert (c4.cpu() - count[:4].cpu()).abs().max().item() <= 5
    for b in range(4):
        if c[b] > 0 and int(c4[b]) > 0:
            assert abs(float(d4[b, 0, 4]) - float(dets[b, 0, 4])) <= 2e-2
    assert c.sum() > 0
    assert m.prefix_status.flagged() == 0   # no image's top-nms_pre cut could have changed its result (as bench.py reports)


def test_benchmark_batch_120_every_chunked_path():
    """bench.py's default shape: R50-FPN 800x1344 at batch 120 in the stem layout -- the P2-level tensors are 4.1 GB, so md_conv2d, md_bottleneck,
    md_conv1x1_dual and md_conv2d_head each run as TWO image chunks, and the stream / persistent / dual ping-pong kernels run at the
    benchmark's grids.  Images 60..119 are copies of images 0..59, so image i sits in the first chunk and its copy in the second: every
    output of the pair must be BIT-IDENTICAL (tile position, chunk and workgroup round cannot leak into a result).  Sampled images:
    proposals and second stage bit-exact against the oracle from the device tensors at the stage boundary; prefix flags clear."""
    from minddet.models import Config, build_detector
    from minddet_amd import nn_ops
    from minddet_amd.data import synthetic_images
    from tests import stage_checks

    cfg = Config.fromfile("configs/faster_rcnn/faster_rcnn_r50_fpn.py")
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    half = nn_ops.to_stem_layout(synthetic_images(60, 800, 1344, seed=20240317, device=DEV))
    x = torch.cat([half, half], 0).contiguous()
    del half
    dets, count, aux = m.forward(x, return_aux=True)
    dets2, count2 = m.forward(x)
    torch.cuda.synchronize()
    assert dets.shape == (120, 100, 6) and count.shape == (120,)
    assert torch.equal(dets, dets2) and torch.equal(count, count2)                       # deterministic
    assert torch.equal(dets[:60], dets[60:]) and torch.equal(count[:60], count[60:])     # chunk / tile position independent
    for f in aux["feats"]:
        assert torch.equal(f[:60], f[60:])
    assert torch.equal(aux["roi"]["pooled"][:60000], aux["roi"]["pooled"][60000:])
    d, c = stage_checks.structure(dets, count, 100, (800, 1344))
    assert c.sum() > 0
    stage_checks.rpn_images(m, aux, (0, 59, 77, 119))
    stage_checks.roi_images(m, aux, dets, count, (800, 1344), (0, 59, 77, 119))
    assert m.prefix_status.flagged() == 0
    del aux
    # against the first 60 images alone (a batch of its own: other grids, other dispatcher choices on the unchunked layers -> last-bit
    # bf16 differences upstream of every index): counts within 2, top scores within 1e-2 (measured r03: printed below)
    d60, c60 = m.forward(x[:60].contiguous())
    torch.cuda.synchronize()
    cdiff = (c60.cpu() - count[:60].cpu()).abs().max().item()
    both = (c60 > 0) & (count[:60] > 0)
    sdiff = (d60[:, 0, 4] - dets[:60, 0, 4])[both].abs().max().item() if bool(both.any()) else 0.0
    print("batch 60 alone vs inside the batch of 120: max count difference", cdiff, "max top-score difference", sdiff)
    assert cdiff == 0 and sdiff <= 1e-3   # measured r03: 0 and 0.0


def test_fused_block_and_dual_gemm_against_the_layer_by_layer_path():
    """R50 backbone features with md_bottleneck + md_conv1x1_dual (the default) against the same weights layer by layer (FUSE_BLOCKS /
    FUSE_DUAL off): md_bottleneck is bit-identical by construction; md_conv1x1_dual rounds the sum of conv3 and the downsample conv to
    bf16 ONCE where the two-launch path rounds three times -> toleranced (rms of the difference <= 1 % of the feature rms per level).
    Also the ADVICE r02 regression: after a weight change + .to() the fused packs must be rebuilt (they hold copies)."""
    from minddet_amd import graphs, nn_ops

    bb = graphs.ResNet(depth=50, seed=3).to(DEV)
    g = torch.Generator().manual_seed(2)
    x8 = torch.zeros((2, 128, 192, 8))
    x8[..., :3] = torch.randn((2, 128, 192, 3), generator=g)
    x = nn_ops.to_stem_layout(x8.to(torch.bfloat16).to(DEV))

    def run(fuse):
        old = graphs.FUSE_BLOCKS, graphs.FUSE_DUAL
        graphs.FUSE_BLOCKS = graphs.FUSE_DUAL = fuse
        try:
            return [f.float() for f in bb(x)]
        finally:
            graphs.FUSE_BLOCKS, graphs.FUSE_DUAL = old

    fused, plain = run(True), run(False)
    assert torch.equal(fused[0], plain[0])          # stage 1: md_bottleneck only, bit-identical
    for a, b in zip(fused[1:], plain[1:]):
        rms = b.pow(2).mean().sqrt().item()
        assert (a - b).pow(2).mean().sqrt().item() <= 1e-2 * rms
    # change the weights of a stage-1 block and of a stage-2 first block, re-pack with .to(): the fused launches must see them
    bb.stages[0][1].conv3.weight = bb.stages[0][1].conv3.weight * 0.5
    bb.stages[1][0].downsample.weight = bb.stages[1][0].downsample.weight * 0.5
    bb.to(DEV)
    fused2, plain2 = run(True), run(False)
    assert not torch.equal(fused2[0], fused[0]) and torch.equal(fused2[0], plain2[0])
    for a, b in zip(fused2[1:], plain2[1:]):
        assert (a - b).pow(2).mean().sqrt().item() <= 1e-2 * b.pow(2).mean().sqrt().item()
