"""CPU: COCO result format + bbox AP (minddet_amd/coco_eval.py) on hand-computed known answers (pycocotools is absent:
parity with it is unpinned; the protocol is the published COCOeval one the reference calls at centernet/eval.py:181-187)."""
import numpy as np

from minddet_amd.coco_eval import COCOBboxEval, convert_eval_format, dets_to_coco


def gt(img, cat, box, **kw):
    return dict(image_id=img, category_id=cat, bbox=list(box), **kw)


def dt(img, cat, box, score):
    return dict(image_id=img, category_id=cat, bbox=list(box), score=score)


def test_convert_eval_format_matches_the_reference_rounding():
    dets = {1: np.array([[10.123, 20.456, 110.129, 220.451, 0.98765]]), 3: np.array([[0, 0, 5.555, 6.666, 0.5]])}
    out = convert_eval_format(dets, 42, [1, 2, 3])
    assert out["images"] == [{"id": 42}]
    assert out["annotations"][0] == {"image_id": 42, "category_id": 1, "bbox": [10.12, 20.46, 100.01, 200.0], "score": 0.99}
    assert out["annotations"][1]["category_id"] == 3 and out["annotations"][1]["bbox"] == [0.0, 0.0, 5.55, 6.67]
    res = dets_to_coco(np.array([[[1, 2, 11, 22, 0.9, 2], [0, 0, 0, 0, 0, 0]]]), np.array([1]), [7], [1, 2, 3])
    assert res == [{"image_id": 7, "category_id": 3, "bbox": [1.0, 2.0, 10.0, 20.0], "score": 0.9}]


def test_perfect_detections_give_ap_one():
    gts = [gt(1, 1, (10, 10, 50, 50)), gt(1, 2, (100, 100, 20, 20)), gt(2, 1, (0, 0, 200, 200))]
    dts = [dt(g["image_id"], g["category_id"], g["bbox"], 0.9) for g in gts]
    s = COCOBboxEval(gts, dts).summarize()
    for k in ("AP", "AP50", "AP75", "AR100", "APs", "APm", "APl"):   # 400 px^2 small, 2500 medium, 40000 large
        assert abs(s[k] - 1.0) < 1e-9, k


def test_false_positive_ranked_first_halves_the_precision():
    gts = [gt(1, 1, (10, 10, 50, 50))]
    dts = [dt(1, 1, (300, 300, 50, 50), 0.9), dt(1, 1, (10, 10, 50, 50), 0.8)]
    s = COCOBboxEval(gts, dts).summarize()
    assert abs(s["AP"] - 0.5) < 1e-9 and abs(s["AP50"] - 0.5) < 1e-9 and s["AR1"] == 0.0 and s["AR10"] == 1.0


def test_iou_thresholds_and_area_ranges():
    gts = [gt(1, 1, (0, 0, 100, 100))]                      # area 10000: large
    dts = [dt(1, 1, (0, 0, 100, 62), 0.9)]                  # IoU 0.62: a match at 0.50, 0.55, 0.60 only; area 6200: medium
    ev = COCOBboxEval(gts, dts)
    s = ev.summarize()
    assert abs(s["AP"] - 0.3) < 1e-9 and abs(s["AP50"] - 1.0) < 1e-9 and s["AP75"] == 0.0
    assert abs(s["APl"] - 0.3) < 1e-9 and s["APm"] == -1.0 and s["APs"] == -1.0
    assert abs(s["AR100"] - 0.3) < 1e-9


def test_crowd_region_absorbs_extra_detections():
    gts = [gt(1, 1, (0, 0, 40, 40)), gt(1, 1, (200, 200, 300, 300), iscrowd=1)]
    dts = [dt(1, 1, (0, 0, 40, 40), 0.9), dt(1, 1, (250, 250, 50, 50), 0.95), dt(1, 1, (300, 300, 60, 60), 0.85)]
    s = COCOBboxEval(gts, dts).summarize()
    assert abs(s["AP"] - 1.0) < 1e-9 and s["AR100"] == 1.0  # both crowd hits are ignored, not false positives


def test_max_dets_limits_recall():
    gts = [gt(1, 1, (0, 0, 40, 40)), gt(1, 1, (100, 100, 40, 40))]
    dts = [dt(1, 1, (0, 0, 40, 40), 0.9), dt(1, 1, (100, 100, 40, 40), 0.8)]
    s = COCOBboxEval(gts, dts).summarize()
    assert s["AR1"] == 0.5 and s["AR10"] == 1.0 and abs(s["AP"] - 1.0) < 1e-9


def test_two_images_interleaved_scores():
    # image 1: TP 0.9 ; image 2: FP 0.8, TP 0.7 -> ranked TP, FP, TP: precision 1, 1/2, 2/3 -> envelope 1, 2/3, 2/3
    gts = [gt(1, 1, (0, 0, 40, 40)), gt(2, 1, (0, 0, 40, 40))]
    dts = [dt(1, 1, (0, 0, 40, 40), 0.9), dt(2, 1, (500, 500, 40, 40), 0.8), dt(2, 1, (0, 0, 40, 40), 0.7)]
    s = COCOBboxEval(gts, dts).summarize()
    want = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101.0            # recall thresholds 0..0.5 -> 1.0 ; 0.51..1.0 -> 2/3
    assert abs(s["AP"] - want) < 1e-6
