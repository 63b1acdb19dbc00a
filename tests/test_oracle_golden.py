"""The CPU oracle against the golden vectors the REFERENCE produced (tests/golden/gen_golden.py).

These pin the oracle; the -m gpu tests then compare the HIP path with the oracle."""
import os

import numpy as np
import pytest

import oracle
from oracle import np_ops


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_rot_nms_aot_matches_reference_operator(golden, tag):
    boxes, thr = golden[f"rotnms_{tag}_boxes"], float(golden[f"rotnms_{tag}_thr"])
    keep, num = oracle.nms_rot_aot(boxes, thr)
    assert num == int(golden[f"rotnms_{tag}_num"])
    np.testing.assert_array_equal(keep, golden[f"rotnms_{tag}_keep"])  # bit-exact indices, zero padded


def test_rot_iou_matrix_matches_reference(golden):
    got = oracle.boxes_iou_bev(golden["ioubev_a"], golden["ioubev_b"])
    ref = golden["ioubev_out"]
    # libm float trig (reference) vs double-rounded trig (oracle): <= 2 ulp of 1.0
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-6)
    assert ((got > 0) == (ref > 0)).all()
    assert abs(got[5, 3] - 1.0) < 1e-5


@pytest.mark.parametrize("eps,key", [(0.0, "ioujit_out_eps0"), (1.0, "ioujit_out_eps1")])
def test_iou_jit_bit_exact(golden, eps, key):
    for fn in (np_ops.iou_jit, oracle.iou_aligned):
        got = fn(golden["ioujit_boxes"], golden["ioujit_query"], eps)
        np.testing.assert_array_equal(got, golden[key])


@pytest.mark.parametrize("thr", [0.01, 0.5, 0.7])
def test_nms_jit_keep_exact(golden, thr):
    dets = golden["nmsjit_dets"]
    ref_keep = golden[f"nmsjit_keep_{thr}"]
    assert np_ops.nms_jit(dets, thr, 0.0) == ref_keep.tolist()
    # C oracle on pre-sorted boxes returns a mask over the sorted order
    order = dets[:, 4].argsort()[::-1]
    mask = oracle.nms_aligned(dets[order, :4], thr, 0.0, mode=0)
    np.testing.assert_array_equal(order[mask.astype(bool)], ref_keep)


def test_nms_jit_eps1(golden):
    dets = golden["nmsjit_dets"]
    order = dets[:, 4].argsort()[::-1]
    mask = oracle.nms_aligned(dets[order, :4], 0.5, 1.0, mode=0)
    np.testing.assert_array_equal(order[mask.astype(bool)], golden["nmsjit_keep_0.5_eps1"])


def test_apply_nms_plus1(golden):
    dets = golden["nmsjit_dets"]
    ref_keep = golden["applynms_keep"]
    got = np_ops.apply_nms(dets[:, [1, 0, 3, 2]], dets[:, 4], 0.5, 100)
    np.testing.assert_array_equal(got, ref_keep)
    order = dets[:, 4].argsort()[::-1]
    mask = oracle.nms_aligned(dets[order, :4], 0.5, 0.0, mode=1).astype(bool)
    np.testing.assert_array_equal(order[mask][:100], ref_keep)


def test_near_bbox_corners_standup(golden):
    rb = golden["near_in"]
    np.testing.assert_array_equal(np_ops.rbbox2d_to_near_bbox(rb), golden["near_out"])
    c = np_ops.center_to_corner_box2d(rb[:, :2], rb[:, 2:4], rb[:, 4])
    np.testing.assert_allclose(c, golden["corners_out"], rtol=0, atol=1e-5)
    np.testing.assert_allclose(np_ops.corner_to_standup_nd(c), golden["standup_out"], rtol=0, atol=1e-5)


def test_box_codec(golden):
    enc = np_ops.second_box_encode(golden["codec_boxes"], golden["codec_anchors"])
    np.testing.assert_array_equal(enc, golden["codec_enc"])
    dec = np_ops.second_box_decode(golden["codec_enc"], golden["codec_anchors"])
    np.testing.assert_array_equal(dec, golden["codec_dec"])
    np.testing.assert_allclose(dec, golden["codec_boxes"], atol=2e-4)  # round trip


def test_anchor_generation(golden):
    car = dict(sizes=[1.6, 3.9, 1.56], anchor_strides=[0.32, 0.32, 0.0], anchor_offsets=[0.16, -39.52, -1.78],
               rotations=[0, 1.57], anchor_range=[0, -39.68, -3, 69.12, 39.68, 1])
    small = np_ops.create_anchors_3d_stride([1, 31, 27], **car)
    np.testing.assert_array_equal(small, golden["anchors_stride_small"])
    full = np_ops.create_anchors_3d_stride([1, 248, 216], **car)
    assert list(full.shape) == golden["anchors_stride_full_shape"].tolist()
    flat = full.reshape(-1, 7)
    np.testing.assert_array_equal(flat[golden["anchors_stride_full_sample_idx"]], golden["anchors_stride_full_sample"])
    np.testing.assert_allclose(flat.astype(np.float64).sum(0), golden["anchors_stride_full_sum64"], rtol=1e-12)


def test_anchor_mask(golden):
    vs = np.array([0.16, 0.16, 4.0], np.float32)
    pcr = np.array([0, -39.68, -3, 69.12, 39.68, 1], np.float32)
    area, mask = np_ops.anchors_mask(golden["amask_coors"], (432, 496), golden["amask_anchors_bv"], vs, pcr, 1)
    np.testing.assert_array_equal(area, golden["amask_area"])
    assert mask.sum() > 0


RANGE_CASES = {
    "a": dict(feature_size=[1, 31, 27], anchor_range=[0.0, -39.68, -1.78, 69.12, 39.68, -1.78], sizes=[1.6, 3.9, 1.56], rotations=[0, 1.57]),
    "b": dict(feature_size=[2, 7, 5], anchor_range=[0.1, -3.3, -3.0, 7.7, 3.3, 1.0], sizes=[[0.6, 1.76, 1.73], [0.6, 0.8, 1.73]], rotations=[0, 0.785, 1.57]),
    "c": dict(feature_size=[1, 248, 216], anchor_range=[0.0, -39.68, -1.0, 69.12, 39.68, -1.0], sizes=[1.6, 3.9, 1.56], rotations=[0, np.pi / 2]),
    "d": dict(feature_size=[1, 1, 4], anchor_range=[2.0, 5.0, 0.5, 2.0, 6.0, 0.5], sizes=[1.0, 2.0, 3.0], rotations=[0.0]),
}
PED_CYCLE = [
    dict(sizes=[0.6, 1.76, 1.73], anchor_strides=[0.16, 0.16, 0.0], anchor_offsets=[0.08, -19.76, -1.465], rotations=[0, 1.57],
         anchor_range=[0, -19.84, -2.5, 47.36, 19.84, 0.5], match_threshold=0.5, unmatch_threshold=0.35),
    dict(sizes=[0.6, 0.8, 1.73], anchor_strides=[0.16, 0.16, 0.0], anchor_offsets=[0.08, -19.76, -1.2], rotations=[0, 1.57],
         anchor_range=[0, -19.84, -2.5, 47.36, 19.84, 0.5], match_threshold=0.45, unmatch_threshold=0.3),
]


def anchors_golden():
    return dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "anchors_range_vectors.npz"), allow_pickle=False))


def test_anchor_range_equals_the_reference():
    """create_anchors_3d_range restated (np_ops, linspace_mode 0) == the reference's own function run on this image's numpy
    (tests/golden/gen_anchors.py), bit for bit; mode 1 (numpy 1.21 arithmetic, parity unpinned) stays within an ulp of the range's scale."""
    g = anchors_golden()
    for tag in ("a", "b", "d"):
        a = np_ops.create_anchors_3d_range(**RANGE_CASES[tag])
        np.testing.assert_array_equal(a, g[f"range_{tag}"])
        a1 = np_ops.create_anchors_3d_range(**RANGE_CASES[tag], linspace_mode=1)
        assert (np.abs(a1.astype(np.float64) - a) <= 2.0 ** -22 * np.maximum(np.abs(a), 64.0)).all()   # an ulp of the range's largest centre
    c = np_ops.create_anchors_3d_range(**RANGE_CASES["c"])
    assert list(c.shape) == g["range_c_shape"].tolist()
    flat = c.reshape(-1, 7)
    np.testing.assert_array_equal(flat[::997], g["range_c_sample"])
    np.testing.assert_allclose(flat.astype(np.float64).sum(0), g["range_c_sum64"], rtol=1e-12)
    # the restated linspace is numpy's own on this image
    for lo, hi, n in ((0.0, 69.12, 216), (-39.68, 39.68, 248), (0.1, 7.7, 5), (2.0, 2.0, 4), (3.0, 9.0, 1)):
        np.testing.assert_array_equal(np_ops.linspace_f32(lo, hi, n, 0), np.linspace(np.float32(lo), np.float32(hi), n, dtype=np.float32))


def test_anchor_range_known_answer():
    a = np_ops.create_anchors_3d_range([1, 4, 5], [0, -2, -1, 4, 2, -1], sizes=[1, 2, 3], rotations=[0, 1.57])
    assert a.shape == (1, 4, 5, 1, 2, 7)
    np.testing.assert_allclose(a[0, 0, :, 0, 0, 0], np.linspace(0, 4, 5), rtol=1e-6)
    np.testing.assert_allclose(a[0, :, 0, 0, 0, 1], np.linspace(-2, 2, 4), rtol=1e-6)
    np.testing.assert_allclose(a[0, 1, 2, 0, 1], [2, -2 + 4 / 3, -1, 1, 2, 3, 1.57], rtol=1e-6)


def test_two_generator_anchor_table_equals_the_reference():
    """TargetAssigner.generate_anchors over the cyclist + pedestrian stride generators (target_assigner.py:227-249)."""
    g = anchors_golden()
    r = np_ops.generate_anchors(PED_CYCLE, [1, 13, 17])
    np.testing.assert_array_equal(r["anchors"], g["concat_small_anchors"])
    np.testing.assert_array_equal(r["matched_thresholds"], g["concat_small_matched"])
    np.testing.assert_array_equal(r["unmatched_thresholds"], g["concat_small_unmatched"])
    f = np_ops.generate_anchors(PED_CYCLE, [1, 248, 296])
    assert list(f["anchors"].shape) == g["concat_full_shape"].tolist()
    flat = f["anchors"].reshape(-1, 7)
    np.testing.assert_array_equal(flat[::1009], g["concat_full_sample"])
    np.testing.assert_allclose(flat.astype(np.float64).sum(0), g["concat_full_sum64"], rtol=1e-12)
    np.testing.assert_array_equal(f["matched_thresholds"][::1009], g["concat_full_matched_sample"])


def test_known_answer_shapes_from_reference_comments():
    # CN/src/predict_by_feat.py:148-150: input (448,672) -> heat (1,80,112,168), topk 100
    rng = np.random.default_rng(3)
    hm = np_ops.sigmoid_clip(rng.normal(-3, 1, (1, 80, 28, 42)).astype(np.float32))
    det, inds, cls = np_ops.centernet_decode(hm, rng.normal(4, 1, (1, 2, 28, 42)).astype(np.float32),
                                             rng.uniform(0, 1, (1, 2, 28, 42)).astype(np.float32))
    assert det.shape == (1, 100, 6) and (np.diff(det[0, :, 4]) <= 0).all()
    assert ((cls >= 0) & (cls < 80)).all() and (inds < 28 * 42).all()


def test_soft_nms_known_answers():
    # two identical boxes: the second is decayed by exp(-1/0.5) = e^-2; a disjoint box is untouched
    d = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [50, 50, 59, 59, 0.7]], np.float32)
    n = np_ops.soft_nms(d, method=2)
    assert n == 3
    np.testing.assert_allclose(sorted(d[:, 4]), sorted([0.9, 0.8 * np.exp(-2.0), 0.7]), rtol=1e-6)
    # hard variant removes it (weight 0 -> score 0 < threshold)
    d = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 9, 0.8], [50, 50, 59, 59, 0.7]], np.float32)
    assert np_ops.soft_nms(d, method=3) == 2
