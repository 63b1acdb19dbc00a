"""CPU: KITTI result format + AP protocol (minddet_amd/kitti_eval.py).

Pinned by the reference where it can run here: the image-bbox protocol of pointpillars/src/core/eval_utils.py (golden vectors
made by tests/golden/gen_kitti_eval.py) -- see test_ms_protocol_equals_the_reference_outputs.  pointpillars/eval_gpu/eval.py
(BEV / 3D / AOS) needs numba.cuda: those parts are checked against (a) a scalar, detection-by-detection restatement of the matching
rules written for this test, (b) hand-computed answers -- parity unpinned there.  Rotated overlaps are injected from the CPU oracle here (the product default is the HIP kernel: see
tests/test_kitti_eval_gpu.py)."""
import numpy as np
import pytest

import oracle
from minddet_amd import kitti_eval as ke


def _rot(b, q, criterion):
    return oracle.rotate_iou_eval(np.ascontiguousarray(b, np.float32), np.ascontiguousarray(q, np.float32), criterion)


def _scalar_statistics(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap, thresh, compute_fp, compute_aos):
    """Sequential scan, one detection at a time, following the rules of eval_gpu/eval.py:166-297."""
    D, G = len(dt_datas), len(gt_datas)
    score = dt_datas[:, -1]
    taken = [False] * D
    low = [bool(compute_fp and score[j] < thresh) for j in range(D)]
    tp = fp = fn = 0
    sim = 0
    matched, deltas = [], []
    for i in range(G):
        if ignored_gt[i] == -1:
            continue
        best, best_val, best_ov, best_is_ignored = -1, None, 0.0, False
        for j in range(D):
            if ignored_det[j] == -1 or taken[j] or low[j] or not overlaps[j, i] > min_overlap:
                continue
            if not compute_fp:
                if best_val is None or score[j] > best_val:
                    best, best_val = j, score[j]
            elif ignored_det[j] == 0 and (overlaps[j, i] > best_ov or best_is_ignored):
                best, best_val, best_ov, best_is_ignored = j, 1, overlaps[j, i], False
            elif ignored_det[j] == 1 and best_val is None:
                best, best_val, best_is_ignored = j, 1, True
        if best_val is None:
            fn += ignored_gt[i] == 0
        elif ignored_gt[i] == 1 or ignored_det[best] == 1:
            taken[best] = True
        else:
            tp += 1
            matched.append(score[best])
            deltas.append(gt_datas[i, 4] - dt_datas[best, 4])
            taken[best] = True
    if compute_fp:
        fp = sum(1 for j in range(D) if not (taken[j] or ignored_det[j] in (-1, 1) or low[j]))
        if metric == 0:
            ov = ke.image_box_overlap(dt_datas[:, :4], dc_bboxes, 0)
            for i in range(len(dc_bboxes)):
                for j in range(D):
                    if taken[j] or ignored_det[j] in (-1, 1) or low[j]:
                        continue
                    if ov[j, i] > min_overlap:
                        taken[j] = True
                        fp -= 1
        if compute_aos:
            sim = sum((1 + np.cos(d)) / 2 for d in deltas) if (tp > 0 or fp > 0) else -1
    return tp, fp, int(fn), sim, np.array(matched)


@pytest.mark.parametrize("seed", range(12))
def test_matching_equals_the_sequential_scan(seed):
    rng = np.random.default_rng(seed)
    D, G, C = int(rng.integers(0, 14)), int(rng.integers(0, 9)), int(rng.integers(0, 3))
    overlaps = np.round(rng.random((D, G)), 1)                    # coarse values: ties in overlap happen
    gt = np.concatenate([rng.random((G, 4)) * 100, rng.uniform(-3, 3, (G, 1))], 1)
    dt = np.concatenate([np.sort(rng.random((D, 4)) * 300, 1)[:, [0, 1, 2, 3]], rng.uniform(-3, 3, (D, 1)), np.round(rng.random((D, 1)), 1)], 1)
    dt[:, :4] = dt[:, [0, 1, 2, 3]]
    ig, idt = rng.integers(-1, 2, G), rng.integers(-1, 2, D)
    dc = np.sort(rng.random((C, 4)) * 300, 1)
    for compute_fp in (False, True):
        for thresh in (0.0, 0.35):
            for metric in (0, 1):
                a = ke.compute_statistics(overlaps, gt, dt, ig, idt, dc, metric, 0.45, thresh, compute_fp, True)
                b = _scalar_statistics(overlaps, gt, dt, ig, idt, dc, metric, 0.45, thresh, compute_fp, True)
                assert a[:3] == b[:3]
                assert a[3] == pytest.approx(b[3])
                assert np.array_equal(a[4], b[4])


def test_thresholds_known_answer():
    assert ke.get_thresholds(np.array([0.8, 0.9]), 2) == [0.9, 0.8]
    # 4 matched of 4: recall 0.25 per detection, sample step 0.025 -> every score is a threshold, highest first
    assert ke.get_thresholds(np.array([0.1, 0.4, 0.3, 0.2]), 4) == [0.4, 0.3, 0.2, 0.1]
    # 100 matched of 100: the 41 sample points pick about every 2.5th score
    t = ke.get_thresholds(np.arange(100) / 100.0, 100)
    assert len(t) == 41 and t[0] == 0.99 and all(a > b for a, b in zip(t, t[1:]))


def test_image_overlap_criteria():
    a = np.array([[0, 0, 10, 10]], float)
    q = np.array([[5, 0, 15, 10], [10, 0, 20, 10], [20, 20, 30, 30]], float)
    assert np.allclose(ke.image_box_overlap(a, q), [[50 / 150, 0, 0]])
    assert np.allclose(ke.image_box_overlap(a, q, 0), [[0.5, 0, 0]])
    assert np.allclose(ke.image_box_overlap(a, q * [1, 1, 2, 1], 1)[0, 0], 50 / 250)
    assert ke.image_box_overlap(np.zeros((0, 4)), q).shape == (0, 3)


def _anno(names, bbox, loc, dims, rot, score=None, occluded=None, truncated=None, alpha=None):
    n = len(names)
    a = dict(name=np.array(names), bbox=np.array(bbox, float).reshape(n, 4), location=np.array(loc, float).reshape(n, 3),
             dimensions=np.array(dims, float).reshape(n, 3), rotation_y=np.array(rot, float).reshape(n),
             occluded=np.array(occluded if occluded is not None else [0] * n), truncated=np.array(truncated if truncated is not None else [0.0] * n),
             alpha=np.array(alpha if alpha is not None else rot, float).reshape(n))
    if score is not None:
        a["score"] = np.array(score, float)
    return a


def _scene():
    gt = [_anno(["Car", "Car", "DontCare"], [[100, 100, 200, 180], [300, 100, 420, 190], [600, 0, 700, 60]],
                [[0, 1.5, 10], [5, 1.5, 20], [-1000, -1000, -1000]], [[4, 1.5, 1.6], [4, 1.5, 1.6], [-1, -1, -1]], [0.0, 0.3, -10]),
          _anno(["Car", "Pedestrian"], [[50, 100, 150, 170], [400, 100, 430, 180]], [[-3, 1.5, 15], [2, 1.6, 8]], [[4, 1.5, 1.6], [0.8, 1.7, 0.6]], [1.2, 0.0])]
    return gt


def test_perfect_detections_score_100_on_every_metric():
    gt = _scene() * 45          # >= 41 objects per class: fewer objects give fewer than 41 thresholds and cap the AP below 100
    dt = []
    rng = np.random.default_rng(0)
    for g in gt:
        keep = g["name"] != "DontCare"
        d = {k: v[keep] for k, v in g.items()}
        d["score"] = rng.uniform(0.3, 0.9, keep.sum())
        # not EXACTLY the ground-truth boxes: the polygon clipper of rotate_iou.py (and its restatements here) can lose the
        # intersection of two identical rotated rectangles to rounding (all edges collinear, corners on the boundary)
        d["location"] = d["location"] + 0.02
        d["dimensions"] = d["dimensions"] * 0.98
        dt.append(d)
    text, out = ke.get_official_eval_result(gt, dt, ["Car", "Pedestrian"], rotate_iou=_rot)
    for k, v in out.items():
        assert v == pytest.approx(100.0), k
    assert "Car AP@ 0.70,  0.70,  0.70:" in text and "bev AP:  100.0000,  100.0000,  100.0000" in text and "3d AP:  100.0000" in text and "aos AP:  100.00" in text
    text_ms, m = ke.get_official_eval_result(gt, dt, [0, 1], protocol="ms")
    assert np.allclose(m, 100.0) and text_ms.startswith("        Easy   Mod    Hard\nCar AP@0.70, 0.70, 0.70:\nbbox AP:  100.00")


def test_hand_computed_average_precision():
    """3 car ground truths (one per image), detections: image 0 a hit (0.9), image 1 a miss-placed box (0.8) plus a hit (0.6),
    image 2 nothing.  Matched scores {0.9, 0.6}, 3 valid gt -> thresholds [0.9, 0.6]; at 0.9: tp 1 fp 0 fn 2; at 0.6: tp 2 fp 1
    fn 1 -> precision [1, 2/3] (envelope keeps it), so the 41-long row is [1, 2/3, 0, ...]: AP11 = (1 + 0 + ...)/11,
    AP_R40 = (2/3)/40."""
    box = [100, 100, 200, 180]
    gts = [_anno(["Car"], [box], [[0, 1.5, 10]], [[4, 1.5, 1.6]], [0.0]) for _ in range(3)]
    hit = dict(bbox=[box], loc=[[0, 1.5, 10]], dims=[[4, 1.5, 1.6]], rot=[0.0])
    far = dict(bbox=[[400, 100, 500, 180]], loc=[[30, 1.5, 40]], dims=[[4, 1.5, 1.6]], rot=[0.0])
    dts = [_anno(["Car"], hit["bbox"], hit["loc"], hit["dims"], hit["rot"], score=[0.9]),
           _anno(["Car", "Car"], far["bbox"] + hit["bbox"], far["loc"] + hit["loc"], far["dims"] + hit["dims"], [0.0, 0.0], score=[0.8, 0.6]),
           dict(ke.empty_result_anno())]
    for metric in (0, 1, 2):
        ret = ke.eval_class(gts, dts, [0], [0], metric, np.full((1, 3, 1), 0.7), rotate_iou=_rot)
        p = ret["precision"][0, 0, 0]
        assert p[0] == pytest.approx(1.0) and p[1] == pytest.approx(2 / 3) and np.all(p[2:] == 0)
        assert ret["recall"][0, 0, 0, :2] == pytest.approx([2 / 3, 2 / 3])      # envelope from the right
        assert ke.get_map(ret["precision"])[0, 0, 0] == pytest.approx(100 / 11)
        assert ke.get_map_r40(ret["precision"])[0, 0, 0] == pytest.approx(100 * (2 / 3) / 40)


def test_difficulty_and_dontcare_rules():
    # gt 0: easy car; gt 1: occluded 2 -> ignored at easy / moderate, counted at hard; a Van is a neighbour class; DontCare region
    gt = [_anno(["Car", "Car", "Van", "DontCare"], [[0, 0, 100, 100], [200, 0, 300, 100], [400, 0, 500, 100], [600, 0, 700, 100]],
                np.zeros((4, 3)), np.ones((4, 3)), np.zeros(4), occluded=[0, 2, 0, -1], truncated=[0, 0, 0, -1])]
    dt = [_anno(["Car", "Car", "Car", "Car", "Car"], [[0, 0, 100, 100], [200, 0, 300, 100], [400, 0, 500, 100], [600, 0, 700, 100], [800, 0, 830, 20]],
                np.zeros((5, 3)), np.ones((5, 3)), np.zeros(5), score=[0.9, 0.8, 0.7, 0.6, 0.5])]
    nv, ig, idt, dc = ke.clean_data(gt[0], dt[0], 0, 0)
    assert (nv, ig, idt, len(dc)) == (1, [0, 1, 1, -1], [0, 0, 0, 0, 1], 1)
    assert ke.clean_data(gt[0], dt[0], 0, 2)[:2] == (2, [0, 0, 1, -1])
    ov = ke.calculate_overlaps(gt, dt, 0)[0]
    gd = np.concatenate([gt[0]["bbox"], gt[0]["alpha"][:, None]], 1)
    dd = np.concatenate([dt[0]["bbox"], dt[0]["alpha"][:, None], dt[0]["score"][:, None]], 1)
    # easy: det 0 true positive; dets 1, 2 sit on ignored ground truth; det 3 lies in the DontCare region; det 4 is too small
    assert ke.compute_statistics(ov, gd, dd, ig, idt, np.array(dc), 0, 0.7, 0.0, True)[:3] == (1, 0, 0)
    # the same detections scored on the BEV metric get no DontCare discount (image metric only)
    assert ke.compute_statistics(ov, gd, dd, ig, idt, np.array(dc), 1, 0.7, 0.0, True)[:3] == (1, 1, 0)


def test_result_format_geometry_and_filters():
    rect, trv2c = np.eye(4), np.eye(4)                        # lidar frame == camera frame for the arithmetic check
    p2 = np.array([[700.0, 0, 600, 0], [0, 700.0, 180, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    boxes = np.array([[0.0, 1.0, 10.0, 1.6, 4.0, 1.5, 0.0], [0.0, 1.0, -5.0, 1.6, 4.0, 1.5, 0.0]])   # (x, y, z, w, l, h, r)
    pred = ke.lidar_boxes_to_prediction(boxes, [0, 0], [0.9, 0.8], rect, trv2c, p2, image_idx=7)
    cam = pred["box3d_camera"][0]
    assert np.allclose(cam, [0, 1, 10, 4.0, 1.5, 1.6, 0])     # (x, y, z, l, h, w, r)
    # corners: x in +-2, y in [1 - 1.5, 1], z in 10 +- 0.8 -> u = 600 + 700 x / z, v = 180 + 700 y / z
    assert np.allclose(pred["bbox"][0], [600 - 700 * 2 / 9.2, 180 + 700 * -0.5 / 9.2, 600 + 700 * 2 / 9.2, 180 + 700 * 1 / 9.2], atol=1e-3)
    annos = ke.predictions_to_kitti_annos([pred, ke.lidar_boxes_to_prediction(None, None, None, rect, trv2c, p2, 8)],
                                          np.array([[375, 1242], [375, 1242]]), ["Car"], center_limit_range=[-10, -10, 0, 10, 10, 70])
    assert list(annos[0]["name"]) == ["Car"] and annos[0]["image_idx"].tolist() == [7]          # the box behind the camera is out of range
    assert annos[0]["alpha"][0] == pytest.approx(-np.arctan2(-1.0, 0.0) + 0.0)
    assert annos[1]["name"].shape == (0,) and annos[1]["bbox"].shape == (0, 4) and annos[1]["image_idx"].shape == (0,)
    line = ke.kitti_result_lines(annos[0])[0].split()
    assert line[0] == "Car" and len(line) == 16 and float(line[8]) == 1.5 and float(line[10]) == 4.0   # h w l order
    # direction classifier flip (predict.py:221-236): rotation > 0 xor dir label -> + pi
    flipped = ke.lidar_boxes_to_prediction(boxes[:1], [0], [0.9], rect, trv2c, p2, 7, dir_labels=[1])
    assert flipped["box3d_lidar"][0, 6] == pytest.approx(np.pi)


# ----------------------------------------------------------------------------- pinned by the reference's own evaluator
def _golden():
    import json
    import os

    return json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "kitti_eval_vectors.json")))


def _np_anno(a):
    return {k: (np.array(v) if k == "name" else np.asarray(v, float if k != "occluded" else int)) for k, v in a.items()}


def test_ms_protocol_equals_the_reference_outputs():
    """tests/golden/kitti_eval_vectors.json = outputs of the reference's pointpillars/src/core/eval_utils.py on synthetic
    annotations (generator: tests/golden/gen_kitti_eval.py): result text, mAP table, the 41-point precision rows for two overlap
    sets, clean_data flags and get_thresholds lists must all be reproduced."""
    gold = _golden()
    for t in gold["thresholds"]:
        assert ke.get_thresholds(np.array(t["scores"]), t["num_gt"]) == pytest.approx(t["thresholds"], abs=0)
    for case in gold["cases"]:
        gt, dt = [_np_anno(a) for a in case["gt"]], [_np_anno(a) for a in case["dt"]]
        text, m = ke.get_official_eval_result(gt, dt, [0, 1, 2], protocol="ms")
        assert text == case["text"]
        np.testing.assert_allclose(m, np.array(case["map_bbox"]), rtol=0, atol=1e-9)
        mo = np.array([[[0.7, 0.5, 0.5]] * 3, [[0.5, 0.25, 0.25]] * 3])
        ret = ke.eval_class(gt, dt, [0, 1, 2], [0, 1, 2], 0, mo, class_names=ke.CLASS_NAMES_MS)
        np.testing.assert_allclose(np.nan_to_num(ret["precision"], nan=-1.0), np.array(case["precision"]), rtol=0, atol=1e-12)
        i = 0
        for g, d in zip(gt[:5], dt[:5]):
            for diff in (0, 2):
                for c in (0, 1):
                    nv, ig, idt, dc = ke.clean_data(g, d, c, diff, ke.CLASS_NAMES_MS)
                    assert [nv, list(ig), list(idt), len(dc)] == case["clean"][i][c]
                i += 1


def test_result_geometry_equals_the_reference_outputs():
    """box_lidar_to_camera / boxes3d_kitti_camera_to_imageboxes of the reference's box_ops.py on a KITTI-like calibration."""
    g = _golden()["geometry"]
    rect, trv2c, p2 = np.array(g["rect"]), np.array(g["trv2c"]), np.array(g["p2"])
    cam = ke.box_lidar_to_camera(np.array(g["boxes_lidar"]), rect, trv2c)
    np.testing.assert_allclose(cam, np.array(g["boxes_camera"]), rtol=0, atol=1e-12)
    img = ke.camera_boxes_to_image_boxes(cam, p2)
    np.testing.assert_allclose(img, np.array(g["boxes_image"]), rtol=1e-6, atol=1e-4)
