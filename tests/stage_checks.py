"""Shared by the full-size -m gpu tests: index-like outputs of SAMPLED images checked bit-exact against the CPU oracle fed the DEVICE
tensors at the stage boundary (the treatment tests/test_yolov5_gpu.py gives YOLOv5s at batch 32), so that a benchmark-size batch is
covered without running the whole oracle on it."""
import numpy as np

from oracle import nets, np_ops


def structure(dets, count, max_det, img_hw):
    """padding, score order, box range of every image; returns (dets, count) as numpy"""
    d, c = dets.cpu().numpy(), count.cpu().numpy()
    H, W = img_hw
    for b in range(d.shape[0]):
        n = c[b]
        assert 0 <= n <= max_det and (np.diff(d[b, :n, 4]) <= 0).all() and (d[b, n:] == 0).all(), b
        assert (d[b, :n, 0] >= 0).all() and (d[b, :n, 1] >= 0).all() and (d[b, :n, 2] <= W).all() and (d[b, :n, 3] <= H).all(), b
        assert (d[b, :n, 2] >= d[b, :n, 0]).all() and (d[b, :n, 3] >= d[b, :n, 1]).all(), b
    return d, c


def rpn_images(model, aux, images):
    """per-level NMS keep masks, merged top-k and RoIs of the sampled images from the DEVICE per-level boxes / scores: bit-exact"""
    rpn = model.rpn_head
    boxes_d, scores_d, counts_d = (aux["rpn"][k].cpu().numpy() for k in ("boxes", "scores", "counts"))
    L, B, k = scores_d.shape
    keep_d = aux["rpn"]["keep"].cpu().numpy().reshape(L, B, k)
    rois_d, cnt_d = aux["rois"].cpu().numpy(), aux["roi_cnt"].cpu().numpy()
    post = rpn.max_per_img
    for b in images:
        keep_o, _, rois_o, cnt_o = nets.proposals_from_lists(boxes_d[:, b:b + 1], scores_d[:, b:b + 1], counts_d[:, b:b + 1], rpn.nms_thr, post)
        np.testing.assert_array_equal(keep_d[:, b], keep_o[:, 0])
        assert cnt_d[b] == cnt_o[0]
        np.testing.assert_array_equal(rois_d[b * post:(b + 1) * post, 1:], rois_o[:, 1:])
        assert (rois_d[b * post:(b + 1) * post, 0] == b).all()


def roi_images(model, aux, dets, count, img_hw, images):
    """second stage of the sampled images from the DEVICE candidate scores / logits / RoIs: selection indices, NMS keep masks, labels,
    scores and counts bit-exact, boxes to fp tolerance; and the prefix flags of every image"""
    roi, post = model.roi_head, model.rpn_head.max_per_img
    d, c = dets.cpu().numpy(), count.cpu().numpy()
    cand_d = aux["roi"]["cand"].cpu().numpy()
    cand_d = np.where(cand_d < -1e30, -np.inf, cand_d)
    rois = aux["rois"].cpu().numpy()
    for b in images:
        cr = aux["roi"]["cls_reg"][b * post:(b + 1) * post].float().cpu().numpy()
        dets_o, count_o, sel = nets.rcnn_finish(cand_d[b:b + 1], cr, rois[b * post:(b + 1) * post], roi.nc, roi.reg_offset, img_hw,
                                                roi.nms_pre, roi.nms_thr, roi.max_per_img, post)
        assert c[b] == count_o[0], b
        np.testing.assert_array_equal(d[b, :, 5], dets_o[0, :, 5])
        np.testing.assert_array_equal(d[b, :, 4], dets_o[0, :, 4])
        np.testing.assert_allclose(d[b, :, :4], dets_o[0, :, :4], rtol=1e-5, atol=2e-3)
        m_ = len(sel[0]["idx"])
        np.testing.assert_array_equal(aux["roi"]["sel_idx"].cpu().numpy()[b, :m_], sel[0]["idx"])
        k_o = sel[0]["keep"] & (np.cumsum(sel[0]["keep"]) <= roi.max_per_img)
        np.testing.assert_array_equal(aux["roi"]["keep"].cpu().numpy()[b, :m_].astype(bool), k_o)
    selc = aux["roi"]["sel_cnt"].cpu().numpy()
    st = roi.prefix_status.tensor(d.shape[0], dets.device).cpu().numpy()
    np.testing.assert_array_equal(st & 1, ((selc >= roi.nms_pre) & (c < roi.max_per_img)).astype(st.dtype))


def one_stage_images(m, aux, dets, count, images, oracle_mod):
    """YOLO: pre-NMS selection, class-aware NMS and packing of the sampled images from the DEVICE decode outputs: bit-exact; prefix flags"""
    d, c = dets.cpu().numpy(), count.cpu().numpy()
    sd, bd, ld = aux["scores"].cpu().numpy(), aux["boxes"].cpu().numpy(), aux["labels"].cpu().numpy()
    for b in images:
        sc = np.where(sd[b] < -1e30, -np.inf, sd[b])
        v, i = np_ops.topk_desc_stable(sc, m.nms_pre)
        k = min(m.nms_pre, int(np.isfinite(sc).sum()))
        v, i = v[:k], i[:k]
        assert int(aux["sel_cnt"][b]) == k
        np.testing.assert_array_equal(aux["sel_idx"].cpu().numpy()[b, :k], i)
        keep = oracle_mod.nms_aligned(bd[b][i], m.iou_thres, 0.0, 2, groups=ld[b][i]).astype(bool)
        kidx = np.nonzero(keep)[0][:m.max_det]
        assert c[b] == len(kidx)
        np.testing.assert_array_equal(d[b, :len(kidx), :4], bd[b][i][kidx])
        np.testing.assert_array_equal(d[b, :len(kidx), 4], v[kidx])
        np.testing.assert_array_equal(d[b, :len(kidx), 5], ld[b][i][kidx])
    st = m.prefix_status.tensor(d.shape[0], dets.device).cpu().numpy()
    selc = aux["sel_cnt"].cpu().numpy()
    np.testing.assert_array_equal(st & 1, ((selc >= m.nms_pre) & (c < m.max_det)).astype(st.dtype))
