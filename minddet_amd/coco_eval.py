"""COCO result format + bbox AP without pycocotools ("the step after the path").

* `convert_eval_format` mirrors minddet/models/centernet/src/post_process.py:64-89: per-class detections -> COCO result
  records (xywh, 2-decimal floats exactly as the reference's `to_float`).
* `dets_to_coco` does the same from the fixed-shape device output `dets [B,max_det,6]` + `count [B]`.
* `COCOBboxEval` restates the published COCOeval bbox protocol the reference calls at centernet/eval.py:181-187
  (`COCOeval(coco, coco_dets, "bbox"); evaluate(); accumulate(); summarize()`): IoU thresholds 0.50:0.05:0.95, 101 recall
  points, area ranges all / small / medium / large, maxDets 1 / 10 / 100, crowd ground truth matched many-to-one with
  IoU = intersection / detection area, ignored ground truth (crowd or outside the area range) sorted last and never
  counted.  pycocotools is not installable here, so the evaluator is checked against hand-computed known answers
  (tests/test_coco_eval_cpu.py): parity with pycocotools itself is unpinned.

Host-side numpy, like the reference's own evaluation step (pycocotools runs on the CPU there too).
"""
import numpy as np


def to_float(x):
    """post_process.py:87-89"""
    return float("{:.2f}".format(x))


def convert_eval_format(detections, img_id, valid_ids):
    """detections: {class index (1-based): array [n, 5] = x1, y1, x2, y2, score} -> {"images": [...], "annotations": [...]}."""
    out = {"images": [], "annotations": []}
    for cls_ind in detections:
        class_id = valid_ids[cls_ind - 1]
        for det in detections[cls_ind]:
            d = np.asarray(det, np.float64)
            bbox = [to_float(d[0]), to_float(d[1]), to_float(d[2] - d[0]), to_float(d[3] - d[1])]
            out["annotations"].append({"image_id": int(img_id), "category_id": int(class_id), "bbox": bbox, "score": to_float(d[4])})
    if out["annotations"]:
        out["images"].append({"id": int(img_id)})
    return out


def dets_to_coco(dets, count, image_ids, valid_ids):
    """dets [B,max_det,6] (x1,y1,x2,y2,score,label 0-based), count [B] -> list of COCO result records."""
    dets, count = np.asarray(dets, np.float64), np.asarray(count)
    res = []
    for b, img_id in enumerate(image_ids):
        per_class = {}
        for d in dets[b, :int(count[b])]:
            per_class.setdefault(int(d[5]) + 1, []).append(d[:5])
        res += convert_eval_format(per_class, img_id, valid_ids)["annotations"]
    return res


def _iou_xywh(dt, gt, iscrowd):
    """[D,4] x [G,4] xywh -> [D,G]; crowd columns use intersection / detection area (pycocotools maskApi.iou semantics)."""
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    dx1, dy1, dx2, dy2 = dt[:, 0:1], dt[:, 1:2], dt[:, 0:1] + dt[:, 2:3], dt[:, 1:2] + dt[:, 3:4]
    gx1, gy1, gx2, gy2 = gt[None, :, 0], gt[None, :, 1], gt[None, :, 0] + gt[None, :, 2], gt[None, :, 1] + gt[None, :, 3]
    iw = np.clip(np.minimum(dx2, gx2) - np.maximum(dx1, gx1), 0, None)
    ih = np.clip(np.minimum(dy2, gy2) - np.maximum(dy1, gy1), 0, None)
    inter = iw * ih
    da = (dt[:, 2] * dt[:, 3])[:, None]
    ga = (gt[:, 2] * gt[:, 3])[None, :]
    union = np.where(np.asarray(iscrowd, bool)[None, :], da, da + ga - inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(union > 0, inter / union, 0.0)


class COCOBboxEval:
    IOU_THRS = np.linspace(0.5, 0.95, 10)
    REC_THRS = np.linspace(0.0, 1.0, 101)
    AREA_RNG = {"all": (0, 1e10), "small": (0, 32 ** 2), "medium": (32 ** 2, 96 ** 2), "large": (96 ** 2, 1e10)}
    MAX_DETS = (1, 10, 100)

    def __init__(self, gt_annotations, dt_results, image_ids=None, category_ids=None):
        """gt_annotations: COCO "annotations" records (image_id, category_id, bbox xywh, [area], [iscrowd], [ignore]);
        dt_results: COCO result records (image_id, category_id, bbox xywh, score)."""
        self.gts, self.dts = {}, {}
        for g in gt_annotations:
            g = dict(g)
            g.setdefault("area", g["bbox"][2] * g["bbox"][3])
            g.setdefault("iscrowd", 0)
            g["_ignore"] = bool(g.get("ignore", 0)) or bool(g["iscrowd"])
            self.gts.setdefault((g["image_id"], g["category_id"]), []).append(g)
        for d in dt_results:
            d = dict(d)
            d.setdefault("area", d["bbox"][2] * d["bbox"][3])
            self.dts.setdefault((d["image_id"], d["category_id"]), []).append(d)
        keys = list(self.gts) + list(self.dts)
        self.img_ids = sorted(set(image_ids if image_ids is not None else [k[0] for k in keys]))
        self.cat_ids = sorted(set(category_ids if category_ids is not None else [k[1] for k in keys]))
        self.precision = self.recall = None

    def _evaluate_img(self, img, cat, rng, max_det):
        gts = self.gts.get((img, cat), [])
        dts = self.dts.get((img, cat), [])
        if not gts and not dts:
            return None
        g_ig = np.array([g["_ignore"] or g["area"] < rng[0] or g["area"] > rng[1] for g in gts], bool)
        g_order = np.argsort(g_ig, kind="mergesort")          # not-ignored ground truth first
        gts = [gts[i] for i in g_order]
        g_ig = g_ig[g_order]
        d_order = np.argsort([-d["score"] for d in dts], kind="mergesort")[:max_det]
        dts = [dts[i] for i in d_order]
        crowd = np.array([g["iscrowd"] for g in gts], bool)
        ious = _iou_xywh(np.array([d["bbox"] for d in dts], float).reshape(-1, 4), np.array([g["bbox"] for g in gts], float).reshape(-1, 4),
                         crowd)
        T, D, G = len(self.IOU_THRS), len(dts), len(gts)
        gtm = -np.ones((T, G), int)
        dtm = -np.ones((T, D), int)
        dt_ig = np.zeros((T, D), bool)
        for ti, t in enumerate(self.IOU_THRS):
            for di in range(D):
                iou, m = min(t, 1 - 1e-10), -1
                for gi in range(G):
                    if gtm[ti, gi] >= 0 and not crowd[gi]:
                        continue                                  # already matched (crowd may match again)
                    if m > -1 and not g_ig[m] and g_ig[gi]:
                        break                                     # a real match was found; only ignored ground truth is left
                    if ious[di, gi] < iou:
                        continue
                    iou, m = ious[di, gi], gi
                if m == -1:
                    continue
                dt_ig[ti, di] = g_ig[m]
                dtm[ti, di] = m
                gtm[ti, m] = di
        d_area = np.array([d["area"] for d in dts])
        out_of_range = (d_area < rng[0]) | (d_area > rng[1])
        dt_ig = dt_ig | ((dtm < 0) & out_of_range[None, :])     # unmatched detections outside the area range are ignored
        return dict(dt_scores=np.array([d["score"] for d in dts]), dtm=dtm, dt_ig=dt_ig, n_gt=int((~g_ig).sum()))

    def evaluate(self):
        T, R, K, A, M = len(self.IOU_THRS), len(self.REC_THRS), len(self.cat_ids), len(self.AREA_RNG), len(self.MAX_DETS)
        self.precision = -np.ones((T, R, K, A, M))
        self.recall = -np.ones((T, K, A, M))
        for k, cat in enumerate(self.cat_ids):
            for a, rng in enumerate(self.AREA_RNG.values()):
                for m, max_det in enumerate(self.MAX_DETS):
                    es = [e for e in (self._evaluate_img(img, cat, rng, max_det) for img in self.img_ids) if e is not None]
                    if not es:
                        continue
                    scores = np.concatenate([e["dt_scores"] for e in es])
                    order = np.argsort(-scores, kind="mergesort")
                    dtm = np.concatenate([e["dtm"] for e in es], 1)[:, order]
                    dig = np.concatenate([e["dt_ig"] for e in es], 1)[:, order]
                    n_gt = sum(e["n_gt"] for e in es)
                    if n_gt == 0:
                        continue
                    tps = np.cumsum((dtm >= 0) & ~dig, 1).astype(float)
                    fps = np.cumsum((dtm < 0) & ~dig, 1).astype(float)
                    for ti in range(T):
                        tp, fp = tps[ti], fps[ti]
                        nd = len(tp)
                        rc = tp / n_gt
                        pr = tp / (fp + tp + np.spacing(1))
                        self.recall[ti, k, a, m] = rc[-1] if nd else 0
                        for i in range(nd - 1, 0, -1):            # precision envelope
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        inds = np.searchsorted(rc, self.REC_THRS, side="left")
                        q = np.zeros(R)
                        valid = inds < nd
                        q[valid] = pr[inds[valid]]
                        self.precision[ti, :, k, a, m] = q
        return self

    def _summarize(self, ap, iou=None, area="all", max_det=100):
        a = list(self.AREA_RNG).index(area)
        m = self.MAX_DETS.index(max_det)
        s = self.precision[:, :, :, a, m] if ap else self.recall[:, :, a, m]
        if iou is not None:
            s = s[np.isclose(self.IOU_THRS, iou)]
        s = s[s > -1]
        return float(s.mean()) if s.size else -1.0

    def summarize(self):
        """The 12 numbers COCOeval.summarize() prints, in its order."""
        if self.precision is None:
            self.evaluate()
        return {
            "AP": self._summarize(True), "AP50": self._summarize(True, 0.5), "AP75": self._summarize(True, 0.75),
            "APs": self._summarize(True, area="small"), "APm": self._summarize(True, area="medium"), "APl": self._summarize(True, area="large"),
            "AR1": self._summarize(False, max_det=1), "AR10": self._summarize(False, max_det=10), "AR100": self._summarize(False),
            "ARs": self._summarize(False, area="small"), "ARm": self._summarize(False, area="medium"), "ARl": self._summarize(False, area="large"),
        }
