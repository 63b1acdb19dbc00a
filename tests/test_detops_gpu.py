"""-m gpu parity: anchors, anchor mask, codecs, top-k, RoIAlign, CenterNet decode, pooling vs the
CPU oracle / golden fixtures.  Bit-exact for indices, anchors and masks; stated fp tolerances else."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


CAR = dict(sizes=[1.6, 3.9, 1.56], anchor_strides=[0.32, 0.32, 0.0], anchor_offsets=[0.16, -39.52, -1.78],
           rotations=[0, 1.57], anchor_range=[0, -39.68, -3, 69.12, 39.68, 1])


def test_anchors_3d_stride_bit_exact(golden):
    from minddet_amd import det_ops

    small = det_ops.create_anchors_3d_stride([1, 31, 27], **CAR).cpu().numpy()
    np.testing.assert_array_equal(small, golden["anchors_stride_small"])  # the reference's own output
    full = det_ops.create_anchors_3d_stride([1, 248, 216], **CAR).cpu().numpy()
    np.testing.assert_array_equal(full, np_ops.create_anchors_3d_stride([1, 248, 216], **CAR))
    flat = full.reshape(-1, 7)
    np.testing.assert_array_equal(flat[golden["anchors_stride_full_sample_idx"]], golden["anchors_stride_full_sample"])
    # pedestrian / cyclist generators (ped_cycle_xyres16.yaml sizes) share the grid
    for size in ([0.6, 1.76, 1.73], [0.6, 0.8, 1.73]):
        cfg = dict(CAR, sizes=size, anchor_range=[0, -19.84, -2.5, 47.36, 19.84, 0.5])
        got = det_ops.create_anchors_3d_stride([1, 248, 296], **cfg).cpu().numpy()
        np.testing.assert_array_equal(got, np_ops.create_anchors_3d_stride([1, 248, 296], **cfg))


def test_anchors_3d_range_and_generator_concat_bit_exact():
    """md_anchors_3d_range == the reference's create_anchors_3d_range (golden, run on this image's numpy = linspace_mode 0) and
    == the oracle in both linspace modes; the cyclist + pedestrian generators written in place into ONE table == the reference's
    TargetAssigner.generate_anchors (values, order and thresholds)."""
    from minddet_amd import det_ops
    from test_oracle_golden import PED_CYCLE, RANGE_CASES, anchors_golden

    g = anchors_golden()
    for tag, kw in RANGE_CASES.items():
        got = det_ops.create_anchors_3d_range(**kw).cpu().numpy()
        np.testing.assert_array_equal(got, np_ops.create_anchors_3d_range(**kw))
        if tag != "c":
            np.testing.assert_array_equal(got, g[f"range_{tag}"])
        else:
            np.testing.assert_array_equal(got.reshape(-1, 7)[::997], g["range_c_sample"])
        got1 = det_ops.create_anchors_3d_range(**kw, linspace_mode=1).cpu().numpy()
        np.testing.assert_array_equal(got1, np_ops.create_anchors_3d_range(**kw, linspace_mode=1))
    gens = [det_ops.AnchorGeneratorStride(**{k: v for k, v in d.items()}) for d in PED_CYCLE]
    r = det_ops.generate_anchors(gens, [1, 13, 17])
    np.testing.assert_array_equal(r["anchors"].cpu().numpy(), g["concat_small_anchors"])
    np.testing.assert_array_equal(r["matched_thresholds"].cpu().numpy(), g["concat_small_matched"])
    np.testing.assert_array_equal(r["unmatched_thresholds"].cpu().numpy(), g["concat_small_unmatched"])
    f = det_ops.generate_anchors(gens, [1, 248, 296])
    flat = f["anchors"].cpu().numpy().reshape(-1, 7)
    assert list(f["anchors"].shape) == g["concat_full_shape"].tolist()
    np.testing.assert_array_equal(flat[::1009], g["concat_full_sample"])
    np.testing.assert_allclose(flat.astype(np.float64).sum(0), g["concat_full_sum64"], rtol=1e-12)
    np.testing.assert_array_equal(f["matched_thresholds"].cpu().numpy()[::1009], g["concat_full_matched_sample"])
    with pytest.raises(Exception, match="rc=2"):   # a generator that does not fit its slot range
        det_ops.create_anchors_3d_stride([1, 13, 17], **{k: v for k, v in PED_CYCLE[0].items() if "threshold" not in k},
                                         out=r["anchors"], slot_off=3)


def test_fpn_anchors_bit_exact_full_size():
    from minddet_amd import det_ops

    sizes = [(200, 336), (100, 168), (50, 84), (25, 42), (13, 21)]
    got = det_ops.fpn_anchors(sizes).cpu().numpy()
    ref = np_ops.fpn_anchors(sizes)
    assert got.shape == (268569, 4)  # SURVEY a6 dagger
    np.testing.assert_array_equal(got, ref)
    got = det_ops.fpn_anchors([(3, 5)], strides=(16,), scale=4.0, ratios=(1.0,)).cpu().numpy()
    np.testing.assert_array_equal(got, np_ops.fpn_anchors([(3, 5)], (16,), 4.0, (1.0,)))
    np.testing.assert_array_equal(got[0], [-32, -32, 32, 32])


def test_anchor_mask_golden_and_full(golden):
    from minddet_amd import det_ops

    vs = np.array([0.16, 0.16, 4.0], np.float32)
    pcr = np.array([0, -39.68, -3, 69.12, 39.68, 1], np.float32)
    area, mask = det_ops.anchors_mask(T(golden["amask_coors"]), (432, 496), T(golden["amask_anchors_bv"]), vs, pcr, 1)
    np.testing.assert_array_equal(area.cpu().numpy(), golden["amask_area"])
    # full anchor set (107136) on a denser cloud, vs the numpy oracle
    rng = np.random.default_rng(2)
    flat = np_ops.create_anchors_3d_stride([1, 248, 216], **CAR).reshape(-1, 7)
    bv = np_ops.rbbox2d_to_near_bbox(flat[:, [0, 1, 3, 4, 6]])
    coors = np.stack([np.zeros(12000, np.int64), rng.integers(0, 496, 12000), rng.integers(0, 432, 12000)], 1).astype(np.int32)
    a_o, m_o = np_ops.anchors_mask(coors, (432, 496), bv, vs, pcr, 1)
    area, mask = det_ops.anchors_mask(T(coors), (432, 496), T(bv), vs, pcr, 1)
    np.testing.assert_array_equal(area.cpu().numpy(), a_o)
    np.testing.assert_array_equal(mask.cpu().numpy(), m_o)
    # empty cloud
    area, mask = det_ops.anchors_mask(torch.zeros((0, 3), dtype=torch.int32, device=DEV), (432, 496), T(bv[:100]), vs, pcr, 1)
    assert float(area.abs().sum()) == 0 and not bool(mask.any())


def test_second_box_decode(golden):
    from minddet_amd import det_ops

    got = det_ops.second_box_decode(T(golden["codec_enc"]), T(golden["codec_anchors"])).cpu().numpy()
    # expf/sqrtf on device vs numpy: <= 2 ulp
    np.testing.assert_allclose(got, golden["codec_dec"], rtol=3e-7, atol=1e-6)
    # batched broadcast [B, A, 7] with A anchors (pointpillars.py:623-652 shape)
    rng = np.random.default_rng(0)
    anc = golden["codec_anchors"]
    enc = rng.normal(0, 0.5, (3, anc.shape[0], 7)).astype(np.float32)
    got = det_ops.second_box_decode(T(enc), T(anc)).cpu().numpy()
    ref = np_ops.second_box_decode(enc, anc[None])
    np.testing.assert_allclose(got, ref, rtol=3e-7, atol=1e-6)


def test_delta2bbox():
    from minddet_amd import det_ops

    rng = np.random.default_rng(5)
    rois = np.concatenate([rng.uniform(0, 600, (5000, 2)), rng.uniform(600, 1300, (5000, 2))], 1).astype(np.float32)
    rois[:, 3] = np.minimum(rois[:, 3], 800)
    d = rng.normal(0, 1.5, (5000, 4)).astype(np.float32)
    got = det_ops.delta2bbox(T(rois), T(d), stds=(0.1, 0.1, 0.2, 0.2), max_shape=(800, 1344)).cpu().numpy()
    ref = np_ops.delta2bbox(rois, d, stds=(0.1, 0.1, 0.2, 0.2), max_shape=(800, 1344))
    np.testing.assert_allclose(got, ref, rtol=1e-6, atol=2e-3)  # expf ulp * up to ~1e3 px
    assert (got[:, 0::2] >= 0).all() and (got[:, 0::2] <= 1344).all()


@pytest.mark.parametrize("n,k", [(5, 10), (1000, 1000), (16384, 1000), (107136, 900), (201600, 1000), (4096, 4096)])
def test_topk_indices_exact(n, k):
    from minddet_amd import det_ops

    rng = np.random.default_rng(n + k)
    L = 3
    s = (1 / (1 + np.exp(-rng.normal(-3, 2, (L, n))))).astype(np.float32)
    s[1] = np.round(s[1] * 50) / 50  # heavy ties -> exercises the stable tie rule
    s[2, : n // 2] = -1.0            # masked-out scores (pointpillars.py:762)
    v, i = det_ops.top_k(T(s), k)
    v, i = v.cpu().numpy(), i.cpu().numpy()
    for l in range(L):
        rv, ri = np_ops.topk_desc_stable(s[l], k)
        m = len(ri)
        np.testing.assert_array_equal(i[l, :m], ri)
        np.testing.assert_array_equal(v[l, :m], rv)


def test_topk_segmented_ragged_and_threshold():
    from minddet_amd import det_ops

    rng = np.random.default_rng(1)
    lens = [0, 7, 1000, 64, 50000]
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    s = rng.uniform(0, 1, off[-1]).astype(np.float32)
    v, i, c = det_ops.topk_segmented(T(s), T(off), 100, min_score=0.05)
    v, i, c = v.cpu().numpy(), i.cpu().numpy(), c.cpu().numpy()
    for l, n in enumerate(lens):
        seg = s[off[l]:off[l + 1]]
        seg_m = np.where(seg > np.float32(0.05), seg, -np.inf)
        rv, ri = np_ops.topk_desc_stable(seg_m, 100)
        m = min(100, int((seg > np.float32(0.05)).sum()))
        assert c[l] == m
        np.testing.assert_array_equal(i[l, :m], ri[:m])
        np.testing.assert_array_equal(v[l, :m], rv[:m])
        assert (i[l, m:] == 0).all()


def test_pp_select_matches_reference_flow():
    """pointpillars.py:753-765: max over classes, mask -> -1, top_k(900)."""
    from minddet_amd import det_ops

    rng = np.random.default_rng(3)
    scores = (1 / (1 + np.exp(-rng.normal(-3, 2, (107136, 1))))).astype(np.float32)
    mask = rng.uniform(0, 1, 107136) < 0.3
    v_o, i_o, _ = np_ops.pp_select(scores, mask, 900)
    top = torch.where(T(mask), T(scores).max(-1)[0], torch.full((107136,), -1.0, device=DEV))
    v, i = det_ops.top_k(top.view(1, -1), 900)
    np.testing.assert_array_equal(i.cpu().numpy()[0], i_o)
    np.testing.assert_array_equal(v.cpu().numpy()[0], v_o)


def test_roi_align_vs_oracle():
    from minddet_amd import det_ops

    rng = np.random.default_rng(13)
    C = 16
    feats = [rng.normal(0, 1, (2, h, w, C)).astype(np.float32) for (h, w) in [(50, 84), (25, 42), (13, 21), (7, 11)]]
    fb = [torch.from_numpy(f).to(torch.bfloat16) for f in feats]
    scales = [1 / 4, 1 / 8, 1 / 16, 1 / 32]
    R = 60
    cx, cy = rng.uniform(0, 336, R), rng.uniform(0, 200, R)
    w, h = np.exp(rng.uniform(np.log(4), np.log(300), R)), np.exp(rng.uniform(np.log(4), np.log(300), R))
    rois = np.stack([rng.integers(0, 2, R), cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
    rois[0, 1:] = [-20, -20, 30, 30]      # partly outside
    rois[1, 1:] = [330, 190, 400, 260]    # mostly outside
    # RoIs exactly ON the level edges (sqrt(wh) = 112, 224, 448: the upper level), one ulp below them, at the clamps
    # (tiny -> P2, huge -> P5, degenerate -> P2)
    edge = []
    for sz in (112.0, 224.0, 448.0):
        edge.append([0, 0.0, 0.0, sz, sz])                                              # on the edge: upper level
        edge.append([1, 0.0, 0.0, sz, np.nextafter(np.float32(sz), np.float32(0))])    # 1 ulp below: still upper (the 1e-6 of the formula)
        edge.append([0, 0.0, 0.0, sz, sz * (1 - 1e-5)])                                 # clearly below: lower level
        edge.append([0, 0.0, 0.0, sz * 2, sz / 2])
    edge += [[0, 5.0, 5.0, 6.0, 6.0], [1, -500.0, -500.0, 2000.0, 2000.0], [0, 50.0, 50.0, 50.0, 80.0], [1, 60.0, 60.0, 40.0, 90.0]]
    rois = np.concatenate([rois, np.asarray(edge, np.float32)], 0)
    R = rois.shape[0]
    out, lv = det_ops.roi_align([f.to(DEV) for f in fb], T(rois), 7, scales, 2, True, return_levels=True)
    out, lv = out.float().cpu().numpy(), lv.cpu().numpy()
    lv_o = np_ops.fpn_level(rois[:, 1:])
    np.testing.assert_array_equal(lv, lv_o)   # integer work: exact (threshold comparisons on both sides)
    assert lv_o[60:72].tolist() == [3, 3, 2, 3, 4, 4, 3, 4, 5, 5, 4, 5] and lv_o[72:].tolist() == [2, 5, 2, 2]
    # the threshold form is the textbook floor(4 + log2(sqrt(wh)/224 + 1e-6)) away from the edges
    far = np.abs(np.log2(np.sqrt(np.maximum((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]), 1e-9)) / 224) % 1.0 - 0.5) < 0.49
    np.testing.assert_array_equal(lv_o[far], np_ops.fpn_level_log2(rois[far, 1:]))
    for r in range(R):
        l = lv[r] - 2
        f = fb[l][int(rois[r, 0])].float().numpy().transpose(2, 0, 1)
        ref = np_ops.roi_align(f, rois[r:r + 1, 1:], 7, scales[l], 2, True)[0].transpose(1, 2, 0)
        # bf16 output rounding (2^-8 rel) + fp32 accumulation order
        np.testing.assert_allclose(out[r], ref, rtol=8e-3, atol=8e-3)


def test_centernet_decode_indices_exact():
    from minddet_amd import det_ops

    rng = np.random.default_rng(8)
    # CN/src/predict_by_feat.py:148-150 shape: input (448,672) -> heat (1,80,112,168)
    for (B, C, H, W) in [(1, 80, 112, 168), (2, 80, 128, 128), (1, 3, 5, 7)]:
        logits = rng.normal(-3, 1.5, (B, C, H, W)).astype(np.float32)
        hm_dev = det_ops.sigmoid_clip(T(logits))
        hm = hm_dev.cpu().numpy()
        np.testing.assert_allclose(hm, np_ops.sigmoid_clip(logits), rtol=2e-6, atol=1e-7)
        wh = rng.uniform(1, 30, (B, 2, H, W)).astype(np.float32)
        reg = rng.uniform(0, 1, (B, 2, H, W)).astype(np.float32)
        K = 100 if H * W >= 100 else 20
        det, inds, cls = det_ops.DetectionDecode(True, K)({"hm": hm_dev, "wh": T(wh), "reg": T(reg)}, return_indices=True)
        d_o, i_o, c_o = np_ops.centernet_decode(hm, wh, reg, K)  # oracle fed the SAME device sigmoid output
        np.testing.assert_array_equal(inds.cpu().numpy(), i_o)
        np.testing.assert_array_equal(cls.cpu().numpy(), c_o)
        np.testing.assert_allclose(det.cpu().numpy(), d_o, rtol=1e-6, atol=1e-5)


def test_maxpool_upsample_slice():
    from minddet_amd import nn_ops

    g = torch.Generator().manual_seed(0)
    x = torch.randn((2, 37, 53, 64), generator=g).to(torch.bfloat16)
    y = nn_ops.maxpool2d(x.to(DEV), 3, 2, 1, zero_pad=True).float().cpu()
    xp = F.pad(x.float().permute(0, 3, 1, 2), (1, 1, 1, 1), value=0.0)
    ref = F.max_pool2d(xp, 3, 2).permute(0, 2, 3, 1)
    assert torch.equal(y, ref)
    y = nn_ops.maxpool2d(x.to(DEV), 3, 2, 1, zero_pad=False).float().cpu()
    assert torch.equal(y, F.max_pool2d(x.float().permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1))
    y = nn_ops.maxpool2d(x.to(DEV), 1, 2, 0, zero_pad=False).float().cpu()
    assert torch.equal(y, x.float()[:, ::2, ::2])
    top = torch.randn((2, 19, 27, 64), generator=g).to(torch.bfloat16)
    y = nn_ops.upsample_add(x.to(DEV), top.to(DEV)).float().cpu()
    up = F.interpolate(top.float().permute(0, 3, 1, 2), size=(37, 53), mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(y, (x.float() + up).to(torch.bfloat16).float())
    s = nn_ops.slice_cast(x.to(DEV), 5, 12).cpu()
    assert torch.equal(s, x.float()[..., 5:17])


def test_standup_boxes_golden(golden):
    from minddet_amd import det_ops

    rb = golden["near_in"]
    got = det_ops.standup_boxes(T(rb)).cpu().numpy()
    np.testing.assert_allclose(got, golden["standup_out"], rtol=0, atol=2e-5)   # the reference's own output
    b7 = np.zeros((rb.shape[0], 7), np.float32)
    b7[:, [0, 1, 3, 4, 6]] = rb
    np.testing.assert_allclose(det_ops.standup_boxes(T(b7)).cpu().numpy(), golden["standup_out"], rtol=0, atol=2e-5)


def test_pp_selected_data_flow():
    """pointpillars.py:753-765 + predict.py:43-98 on device vs numpy (car config: pre 900? uses 1000 here, post 300)."""
    from minddet_amd import det_ops
    import oracle

    rng = np.random.default_rng(5)
    n = 107136
    scores = (1 / (1 + np.exp(-rng.normal(-4, 2, (n, 1))))).astype(np.float32)
    boxes = np.concatenate([rng.uniform(0, 69, (n, 1)), rng.uniform(-39, 39, (n, 1)), rng.uniform(-2, 0, (n, 1)),
                            rng.uniform(1.4, 1.8, (n, 1)), rng.uniform(3.5, 4.5, (n, 1)), rng.uniform(1.4, 1.7, (n, 1)),
                            rng.uniform(-3.14, 3.14, (n, 1))], 1).astype(np.float32)
    amask = rng.uniform(0, 1, n) < 0.4
    cfg = dict(nms_pre_max_size=1000, nms_post_max_size=300, nms_iou_threshold=0.01, nms_score_threshold=0.05)
    b, s, l, cnt = det_ops.pp_get_selected_data(T(scores), T(boxes), T(amask), cfg)
    cnt = int(cnt)
    top = np.where(amask, scores[:, 0], np.float32(-1))
    keepable = top >= np.float32(0.05)
    v, i = np_ops.topk_desc_stable(np.where(keepable, top, -np.inf), 1000)
    m = min(1000, int(keepable.sum()))
    v, i = v[:m], i[:m]
    st = np_ops.corner_to_standup_nd(np_ops.center_to_corner_box2d(boxes[i][:, :2], boxes[i][:, 3:5], boxes[i][:, 6]))
    st_dev = det_ops.standup_boxes(T(boxes[i])).cpu().numpy()
    np.testing.assert_allclose(st_dev, st, atol=2e-5)
    keep = oracle.nms_aligned(st_dev, 0.01, 0.0, 0).astype(bool)   # NMS from the device standup boxes: exact
    kidx = np.nonzero(keep)[0][:300]
    assert cnt == len(kidx)
    np.testing.assert_array_equal(s.cpu().numpy()[:cnt], v[kidx])
    np.testing.assert_array_equal(b.cpu().numpy()[:cnt], boxes[i][kidx])
    assert (l.cpu().numpy()[:cnt] == 0).all()


def test_soft_nms_vs_published_algorithm():
    """md_soft_nms vs the oracle's restatement of the published CenterNet soft_nms (parity unpinned: the Cython
    module is not in the reference).  Compared as sets keyed by original index: survivors and their decayed scores."""
    from minddet_amd import det_ops

    rng = np.random.default_rng(2)
    for n, method in [(100, 2), (37, 2), (100, 1), (1, 2), (300, 2)]:
        cx, cy = rng.uniform(0, 200, n), rng.uniform(0, 150, n)
        w, h = rng.uniform(10, 80, n), rng.uniform(10, 80, n)
        boxes = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1).astype(np.float32)
        scores = (rng.uniform(0.01, 1, n) + np.arange(n) * 1e-6).astype(np.float32)
        so, order, num = det_ops.soft_nms(T(boxes), T(scores), method=method)
        so, order, num = so.cpu().numpy(), order.cpu().numpy(), int(num[0])
        rows = np.concatenate([boxes, scores[:, None]], 1).astype(np.float32)
        tagged = np.concatenate([rows, np.arange(n, dtype=np.float32)[:, None]], 1)   # 6th column: original index
        cnt = np_ops.soft_nms(tagged, method=method)
        surv = {int(r[5]): r[4] for r in tagged[:cnt]}
        got = {i: so[i] for i in range(n) if so[i] > 0}
        assert set(got) == set(surv) and num == cnt
        for i in got:
            assert abs(got[i] - surv[i]) <= 2e-6 * max(1.0, abs(surv[i])), (i, got[i], surv[i])
        assert sorted(order[:num].tolist()) == sorted(surv)


def test_centernet_post_process_merge():
    """post_process + merge_outputs (centernet/src/post_process.py:10-61) on device vs numpy."""
    from minddet_amd import det_ops

    rng = np.random.default_rng(4)
    K, C = 100, 80
    cx, cy = rng.uniform(0, 128, K), rng.uniform(0, 128, K)
    w, h = rng.uniform(4, 40, K), rng.uniform(4, 40, K)
    dets = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, np.sort(rng.uniform(0.02, 0.9, K))[::-1],
                     rng.integers(0, 6, K)], 1).astype(np.float32)
    c, s, out_hw, scale = np.array([320.0, 240.0], np.float32), 640.0, (128, 128), 1.0
    boxes, scores, cls = det_ops.centernet_post_process(T(dets), c, s, out_hw, scale, C, soft=True)
    tr = det_ops.get_affine_transform(c, s, (128, 128))
    # rot = 0: pure scale + shift: feature (64,64) -> image centre
    np.testing.assert_allclose(tr @ np.array([64.0, 64.0, 1.0]), c, atol=1e-4)
    ref_boxes = np.concatenate([dets[:, 0:2].astype(np.float64) @ tr[:, :2].T + tr[:, 2],
                                dets[:, 2:4].astype(np.float64) @ tr[:, :2].T + tr[:, 2]], 1).astype(np.float32) / scale
    surv_scores = np.zeros(K, np.float32)
    for j in range(C):
        idx = np.nonzero(dets[:, 5] == j)[0]
        if len(idx) == 0:
            continue
        tagged = np.concatenate([ref_boxes[idx], dets[idx, 4:5], idx[:, None].astype(np.float32)], 1).astype(np.float32)
        cnt = np_ops.soft_nms(tagged, method=2)
        for r in tagged[:cnt]:
            surv_scores[int(r[5])] = r[4]
    alive = surv_scores > 0
    if alive.sum() > 100:
        sa = surv_scores[alive]
        alive &= surv_scores >= np.partition(sa, len(sa) - 100)[len(sa) - 100]
    np.testing.assert_allclose(boxes.cpu().numpy(), ref_boxes[alive], atol=1e-3)
    np.testing.assert_allclose(scores.cpu().numpy(), surv_scores[alive], rtol=3e-6, atol=1e-7)
    np.testing.assert_array_equal(cls.cpu().numpy(), dets[alive, 5].astype(np.int32))


@pytest.mark.parametrize("shape", [(2, 128, 128, 88, 80), (1, 37, 70, 24, 19), (3, 8, 64, 16, 3), (1, 5, 3, 8, 8)])
def test_heat_peaks_equals_the_three_passes(shape):
    """md_heat_peaks == md_nhwc_to_nchw_f32 -> md_sigmoid_clip -> md_heat_nms, bit for bit (ragged tiles, class counts that are
    not a multiple of the kernel's 16-class groups, plateaus from the clip)."""
    from minddet_amd import _lib, det_ops, nn_ops

    B, H, W, Cp, nc = shape
    g = torch.Generator().manual_seed(H * W)
    head = (torch.randn((B, H, W, Cp), generator=g) * 6).to(torch.bfloat16)      # |x| up to ~20: both clip ends are hit
    head[:, : H // 2, : W // 2, 0] = 3.0                                            # a plateau: every pixel equals its 3x3 max
    hd = head.to(DEV)
    hm_ref = det_ops.sigmoid_clip(nn_ops.nhwc_to_nchw_f32(hd, 0, nc))
    heat_ref = torch.empty_like(hm_ref)
    _lib.call("md_heat_nms", [hm_ref, heat_ref])
    heat, hm = det_ops.heat_peaks(hd, 0, nc, with_hm=True)
    assert torch.equal(hm, hm_ref) and torch.equal(heat, heat_ref)
    heat2, none = det_ops.heat_peaks(hd, 0, nc)
    assert none is None and torch.equal(heat2, heat_ref)
    assert (heat_ref > 0).sum() > 0 and (heat_ref == 0).sum() > 0
    with pytest.raises(_lib.MindDetHipError):
        det_ops.heat_peaks(hd, 4, nc)
    with pytest.raises(_lib.MindDetHipError):
        det_ops.heat_peaks(hd, 0, Cp + 8)
