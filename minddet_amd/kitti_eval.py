"""KITTI result format + AP evaluation (the 3D half: "COCO/KITTI result formats + AP evaluation").

What the reference does after its PointPillars path:
  * pointpillars/src/predict.py:204-262, 331-395 -- lidar boxes -> camera boxes -> image boxes -> KITTI annotation dicts
    (`lidar_boxes_to_prediction`, `predictions_to_kitti_annos` here);
  * pointpillars/src/core/eval_utils.py (what pointpillars/eval.py:149 calls: image-bbox AP only, 11 recall points) and
    pointpillars/eval_gpu/eval.py (the full protocol: bbox / BEV / 3D AP + AOS, 11- and 40-point AP, rotated overlaps from a
    numba-CUDA kernel) -- `get_official_eval_result(..., protocol="ms" | "full")` here.

Protocol (KITTI object benchmark as both files state it): per class and difficulty (min image height 40/25/25 px, max
occlusion 0/1/2, max truncation 0.15/0.3/0.5) ground truth is split into counted / ignored (neighbour class or too hard) /
other; detections below the height limit are ignored.  A first greedy pass (highest score first per ground truth) collects the
scores of all matched detections; 41 score thresholds are picked from them at equally spaced recall; a second pass per
threshold counts TP / FP / FN with the best-overlap assignment, detections matching ignored ground truth or lying in DontCare
regions (image metric only) are not false positives.  precision[i] = max over later thresholds; AP = mean of precision at
recall points 0, 4, ..., 40 (11-point) or 1..40 (R40).  AOS accumulates (1 + cos(delta alpha)) / 2 over true positives.

The rotated BEV intersections come from `det_ops.rotate_iou_gpu_eval` (md_rotate_iou_eval, the HIP counterpart of
eval_gpu/rotate_iou.py:264-340) exactly where the reference calls its CUDA kernel; everything else is host numpy like the
reference's.  Pinning (tests/test_kitti_eval_cpu.py): the "ms" protocol reproduces the reference's own outputs -- result text,
mAP table, 41-point precision rows, clean_data flags, get_thresholds -- on 223 synthetic images (tests/golden/
kitti_eval_vectors.json, written by tests/golden/gen_kitti_eval.py from pointpillars/src/core/eval_utils.py imported with numba's
decorators as identities).  The "full" protocol shares that machinery; its BEV / 3D / AOS parts (eval_gpu/eval.py needs
numba.cuda) are checked against a scalar restatement of the matching rules and hand-computed answers only: parity unpinned there,
as is any run on real KITTI results.  The lidar -> camera -> image box geometry reproduces the reference's box_ops.py outputs (same
golden file); the annotation writer's filter rules (predict.py needs skimage through kitti_common: not importable) are restated.  Reference quirk: eval_utils.get_split_parts has no guard, fewer than 50 images crash it.
"""
import numpy as np

CLASS_NAMES_FULL = ("car", "pedestrian", "cyclist", "van", "person_sitting", "truck")                 # eval_gpu/eval.py:31
CLASS_NAMES_MS = ("car", "pedestrian", "cyclist", "van", "person_sitting", "car", "tractor", "trailer")  # eval_utils.py:98-107
MIN_HEIGHT = (40, 25, 25)
MAX_OCCLUSION = (0, 1, 2)
MAX_TRUNCATION = (0.15, 0.3, 0.5)
N_SAMPLE_PTS = 41
_NO_DETECTION = -10000000


# ----------------------------------------------------------------------------- result format (predict.py)
def lidar_to_camera(points, r_rect, velo2cam):
    """box_ops.py:502-510: [.., 3] lidar points -> rectified camera coordinates."""
    pts = np.concatenate([points, np.ones(list(points.shape[:-1]) + [1])], axis=-1)
    return (pts @ (r_rect @ velo2cam).T)[..., :3]


def box_lidar_to_camera(boxes, r_rect, velo2cam):
    """box_ops.py:538-546: (x, y, z, w, l, h, r) lidar -> (x, y, z, l, h, w, r) camera."""
    xyz = lidar_to_camera(boxes[..., 0:3], r_rect, velo2cam)
    w, l, h, r = boxes[..., 3:4], boxes[..., 4:5], boxes[..., 5:6], boxes[..., 6:7]
    return np.concatenate([xyz, l, h, w, r], axis=-1)


def camera_box_corners(boxes):
    """box_ops.py:550-631: camera boxes [N,7] (bottom-centre origin) -> corners [N,8,3] float32."""
    n = boxes.shape[0]
    l, h, w, ry = boxes[:, 3], boxes[:, 4], boxes[:, 5], boxes[:, 6]
    xc = np.stack([l / 2, l / 2, -l / 2, -l / 2, l / 2, l / 2, -l / 2, -l / 2], 1).astype(np.float32)
    zc = np.stack([w / 2, -w / 2, -w / 2, w / 2, w / 2, -w / 2, -w / 2, w / 2], 1).astype(np.float32)
    yc = np.zeros((n, 8), np.float32)
    yc[:, 4:8] = -h.reshape(n, 1)
    c, s = np.cos(ry), np.sin(ry)
    zeros, ones = np.zeros(n, np.float32), np.ones(n, np.float32)
    rot = np.transpose(np.array([[c, zeros, -s], [zeros, ones, zeros], [s, zeros, c]]), (2, 0, 1))   # [N,3,3]
    corners = np.matmul(np.stack([xc, yc, zc], 2), rot) + boxes[:, None, 0:3]
    return corners.astype(np.float32)


def camera_boxes_to_image_boxes(boxes, p2):
    """box_ops.py:641-668: project the 8 corners with P2 ([4,4] or [3,4]) and take their bounding rectangle [N,4]."""
    corners = camera_box_corners(boxes).reshape(-1, 3)
    hom = np.hstack((corners, np.ones((corners.shape[0], 1), np.float32)))
    uvw = hom @ np.asarray(p2).T
    uv = (uvw[:, 0:2].T / hom[:, 2]).T.reshape(-1, 8, 2)
    return np.concatenate([uv.min(1), uv.max(1)], axis=1)


def lidar_boxes_to_prediction(boxes_lidar, labels, scores, rect, trv2c, p2, image_idx, dir_labels=None):
    """predict.py:204-262 (`generate_single_sample_dict_old`): selected lidar boxes -> the prediction dict the annotation writer
    consumes.  `dir_labels` applies the direction-classifier flip (:221-236)."""
    if boxes_lidar is None or len(boxes_lidar) == 0:
        return dict(bbox=None, box3d_camera=None, box3d_lidar=None, scores=None, label_preds=None, image_idx=image_idx)
    boxes_lidar = np.array(boxes_lidar, np.float64)
    if dir_labels is not None:
        opp = (boxes_lidar[..., -1] > 0) ^ np.asarray(dir_labels).astype(bool)
        boxes_lidar[..., -1] += np.where(opp, np.pi, 0.0)
    cam = box_lidar_to_camera(boxes_lidar, rect, trv2c)
    return dict(bbox=camera_boxes_to_image_boxes(cam, p2), box3d_camera=cam, box3d_lidar=boxes_lidar, scores=np.asarray(scores),
                label_preds=np.asarray(labels), image_idx=image_idx)


def empty_result_anno():
    """kitti_common.py:439-455"""
    return dict(name=np.array([]), truncated=np.array([]), occluded=np.array([]), alpha=np.array([]), bbox=np.zeros([0, 4]),
                dimensions=np.zeros([0, 3]), location=np.zeros([0, 3]), rotation_y=np.array([]), score=np.array([]))


def predictions_to_kitti_annos(predictions, batch_image_shape, class_names, center_limit_range=None, lidar_input=False):
    """predict.py:331-395 (`predict_kitti_to_anno`): drop boxes outside the image / the centre range, clip the image box, write
    (name, truncated 0, occluded 0, alpha, bbox, dimensions, location, rotation_y, score) + image_idx."""
    annos = []
    for i, pred in enumerate(predictions):
        image_shape = np.asarray(batch_image_shape[i])
        rows = {k: [] for k in ("name", "truncated", "occluded", "alpha", "bbox", "dimensions", "location", "rotation_y", "score")}
        if pred["bbox"] is not None:
            for box, box_lidar, bbox, score, label in zip(pred["box3d_camera"], pred["box3d_lidar"], pred["bbox"], pred["scores"],
                                                          pred["label_preds"]):
                bbox = np.array(bbox, np.float64)
                if not lidar_input:
                    if bbox[0] > image_shape[1] or bbox[1] > image_shape[0] or bbox[2] < 0 or bbox[3] < 0:
                        continue
                if center_limit_range is not None:
                    lim = np.asarray(center_limit_range)
                    if np.any(box_lidar[:3] < lim[:3]) or np.any(box_lidar[:3] > lim[3:]):
                        continue
                bbox[2:] = np.minimum(bbox[2:], image_shape[::-1])
                bbox[:2] = np.maximum(bbox[:2], [0, 0])
                rows["name"].append(class_names[int(label)])
                rows["truncated"].append(0.0)
                rows["occluded"].append(0)
                rows["alpha"].append(-np.arctan2(-box_lidar[1], box_lidar[0]) + box[6])
                rows["bbox"].append(bbox)
                rows["dimensions"].append(box[3:6])
                rows["location"].append(box[:3])
                rows["rotation_y"].append(box[6])
                rows["score"].append(score)
        anno = {k: np.stack(v) for k, v in rows.items()} if rows["name"] else empty_result_anno()
        anno["image_idx"] = np.array([pred["image_idx"]] * anno["name"].shape[0], dtype=np.int64)
        annos.append(anno)
    return annos


def kitti_result_lines(anno):
    """One KITTI label-file line per detection: name truncated occluded alpha bbox(4) dimensions as (h, w, l) location(3)
    rotation_y score -- `dimensions` is stored (l, h, w) as above (kitti_common.py convention)."""
    lines = []
    for i in range(len(anno["name"])):
        l, h, w = anno["dimensions"][i]
        vals = [anno["alpha"][i], *anno["bbox"][i], h, w, l, *anno["location"][i], anno["rotation_y"][i], anno["score"][i]]
        lines.append(f"{anno['name'][i]} {float(anno['truncated'][i]):.2f} {int(anno['occluded'][i])} " + " ".join(f"{float(v):.4f}" for v in vals))
    return lines


# ----------------------------------------------------------------------------- overlaps
def image_box_overlap(boxes, query_boxes, criterion=-1):
    """eval_gpu/eval.py:91-121: [N,4] x [K,4] image boxes; criterion -1 IoU, 0 / boxes area, 1 / query area, else intersection."""
    boxes, query_boxes = np.asarray(boxes, np.float64).reshape(-1, 4), np.asarray(query_boxes, np.float64).reshape(-1, 4)
    iw = np.minimum(boxes[:, None, 2], query_boxes[None, :, 2]) - np.maximum(boxes[:, None, 0], query_boxes[None, :, 0])
    ih = np.minimum(boxes[:, None, 3], query_boxes[None, :, 3]) - np.maximum(boxes[:, None, 1], query_boxes[None, :, 1])
    ok = (iw > 0) & (ih > 0)
    inter = np.where(ok, iw * ih, 0.0)
    a = ((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]))[:, None]
    q = ((query_boxes[:, 2] - query_boxes[:, 0]) * (query_boxes[:, 3] - query_boxes[:, 1]))[None, :]
    if criterion == -1:
        ua = a + q - inter
    elif criterion == 0:
        ua = a + 0 * q
    elif criterion == 1:
        ua = q + 0 * a
    else:
        ua = np.ones_like(inter)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where(ok, inter / ua, 0.0)


def _device_rotate_iou(boxes, query_boxes, criterion):
    import torch

    from . import det_ops
    dev = "cuda:0"
    return det_ops.rotate_iou_gpu_eval(torch.from_numpy(np.ascontiguousarray(boxes, np.float32)).to(dev),
                                       torch.from_numpy(np.ascontiguousarray(query_boxes, np.float32)).to(dev), criterion).cpu().numpy()


def bev_box_overlap(boxes, qboxes, criterion=-1, rotate_iou=None):
    """eval_gpu/eval.py:124-126: [N,5] x [K,5] (x, z, l, w, ry)."""
    if len(boxes) == 0 or len(qboxes) == 0:
        return np.zeros((len(boxes), len(qboxes)))
    return np.asarray((rotate_iou or _device_rotate_iou)(boxes, qboxes, criterion), np.float64)


def d3_box_overlap(boxes, qboxes, criterion=-1, rotate_iou=None):
    """eval_gpu/eval.py:129-162: camera boxes [N,7] (x, y, z, l, h, w, ry), y = bottom face, -y up: BEV intersection area x
    overlap of the [y - h, y] intervals over the volume union."""
    if len(boxes) == 0 or len(qboxes) == 0:
        return np.zeros((len(boxes), len(qboxes)))
    rinc = np.asarray((rotate_iou or _device_rotate_iou)(boxes[:, [0, 2, 3, 5, 6]], qboxes[:, [0, 2, 3, 5, 6]], 2), np.float64)
    ih = np.minimum(boxes[:, None, 1], qboxes[None, :, 1]) - np.maximum(boxes[:, None, 1] - boxes[:, None, 4], qboxes[None, :, 1] - qboxes[None, :, 4])
    v1 = (boxes[:, 3] * boxes[:, 4] * boxes[:, 5])[:, None]
    v2 = (qboxes[:, 3] * qboxes[:, 4] * qboxes[:, 5])[None, :]
    inc = ih * rinc
    ok = (rinc > 0) & (ih > 0)
    ua = {-1: v1 + v2 - inc, 0: v1 + 0 * v2, 1: v2 + 0 * v1}.get(criterion, inc)
    with np.errstate(divide="ignore", invalid="ignore"):
        out = np.where(ok, inc / ua, 0.0)
    return np.where(rinc > 0, out, rinc)          # cells without BEV intersection keep the kernel's value (0)


# ----------------------------------------------------------------------------- matching
def get_thresholds(scores, num_gt, num_sample_pts=N_SAMPLE_PTS):
    """eval_gpu/eval.py:9-27: scores of matched detections -> the thresholds at (about) equally spaced recall."""
    scores = np.sort(np.asarray(scores, np.float64))[::-1]
    current_recall, thresholds = 0.0, []
    for i, score in enumerate(scores):
        l_recall = (i + 1) / num_gt
        r_recall = (i + 2) / num_gt if i < len(scores) - 1 else l_recall
        if (r_recall - current_recall) < (current_recall - l_recall) and i < len(scores) - 1:
            continue
        thresholds.append(score)
        current_recall += 1 / (num_sample_pts - 1.0)
    return thresholds


def clean_data(gt_anno, dt_anno, current_class, difficulty, class_names=CLASS_NAMES_FULL):
    """eval_gpu/eval.py:30-87 / eval_utils.py:36-115 -> (num_valid_gt, ignored_gt, ignored_dt, dc_bboxes); flags 0 counted,
    1 ignored, -1 other class."""
    cur = class_names[current_class].lower()
    ignored_gt, ignored_dt, dc = [], [], []
    num_valid = 0
    for i in range(len(gt_anno["name"])):
        bbox = gt_anno["bbox"][i]
        name = str(gt_anno["name"][i]).lower()
        height = bbox[3] - bbox[1]
        if class_names is CLASS_NAMES_MS:
            height = abs(height)                      # eval_utils.py:49 takes the magnitude, eval_gpu/eval.py:44 does not
        if name == cur:
            valid = 1
        elif (cur == "pedestrian" and name == "person_sitting") or (cur == "car" and name == "van"):
            valid = 0
        else:
            valid = -1
        ignore = (gt_anno["occluded"][i] > MAX_OCCLUSION[difficulty] or gt_anno["truncated"][i] > MAX_TRUNCATION[difficulty]
                  or height <= MIN_HEIGHT[difficulty])
        if valid == 1 and not ignore:
            ignored_gt.append(0)
            num_valid += 1
        elif valid == 0 or (ignore and valid == 1):
            ignored_gt.append(1)
        else:
            ignored_gt.append(-1)
        if gt_anno["name"][i] == "DontCare":
            dc.append(bbox)
    for i in range(len(dt_anno["name"])):
        height = abs(dt_anno["bbox"][i, 3] - dt_anno["bbox"][i, 1])
        if height < MIN_HEIGHT[difficulty]:
            ignored_dt.append(1)
        elif str(dt_anno["name"][i]).lower() == cur:
            ignored_dt.append(0)
        else:
            ignored_dt.append(-1)
    return num_valid, ignored_gt, ignored_dt, dc


def compute_statistics(overlaps, gt_datas, dt_datas, ignored_gt, ignored_det, dc_bboxes, metric, min_overlap, thresh=0.0,
                       compute_fp=False, compute_aos=False):
    """eval_gpu/eval.py:166-297.  overlaps [D,G]; gt_datas [G,5] (bbox, alpha); dt_datas [D,6] (bbox, alpha, score).
    The per-ground-truth scan over detections is evaluated with array operations; the outcome per ground truth is the one the
    reference's sequential scan reaches: without compute_fp the highest-scoring candidate (first on ties); with compute_fp the
    counted candidate of largest overlap (first on ties), else the first ignored candidate.
    -> (tp, fp, fn, similarity, matched scores)."""
    ignored_gt, ignored_det = np.asarray(ignored_gt), np.asarray(ignored_det)
    D, G = dt_datas.shape[0], gt_datas.shape[0]
    dt_scores, dt_alphas, gt_alphas = dt_datas[:, -1], dt_datas[:, 4], gt_datas[:, 4]
    assigned = np.zeros(D, bool)
    below = (dt_scores < thresh) if compute_fp else np.zeros(D, bool)
    usable = (ignored_det != -1) & ~below
    tp = fp = fn = 0
    similarity = 0
    matched_scores, delta = [], []
    for i in range(G):
        if ignored_gt[i] == -1:
            continue
        cand = usable & ~assigned & (overlaps[:, i] > min_overlap) if D else np.zeros(0, bool)
        det_idx = -1
        if not compute_fp:
            if cand.any():
                idx = np.flatnonzero(cand)
                det_idx = int(idx[np.argmax(dt_scores[idx])])
        else:
            counted = cand & (ignored_det == 0)
            if counted.any():
                idx = np.flatnonzero(counted)
                det_idx = int(idx[np.argmax(overlaps[idx, i])])
            else:
                ign = cand & (ignored_det == 1)
                if ign.any():
                    det_idx = int(np.flatnonzero(ign)[0])
        if det_idx < 0:
            if ignored_gt[i] == 0:
                fn += 1
        elif ignored_gt[i] == 1 or ignored_det[det_idx] == 1:
            assigned[det_idx] = True
        else:
            tp += 1
            matched_scores.append(dt_scores[det_idx])
            if compute_aos:
                delta.append(gt_alphas[i] - dt_alphas[det_idx])
            assigned[det_idx] = True
    if compute_fp:
        fp = int(np.sum(~(assigned | (ignored_det == -1) | (ignored_det == 1) | below)))
        nstuff = 0
        if metric == 0 and len(dc_bboxes):
            ov = image_box_overlap(dt_datas[:, :4], dc_bboxes, 0)
            for i in range(len(dc_bboxes)):
                hit = ~assigned & (ignored_det == 0) & ~below & (ov[:, i] > min_overlap)
                nstuff += int(hit.sum())
                assigned |= hit
        fp -= nstuff
        if compute_aos:
            similarity = float(np.sum((1.0 + np.cos(np.asarray(delta, np.float64))) / 2.0)) if (tp > 0 or fp > 0) else -1
    return tp, fp, fn, similarity, np.asarray(matched_scores, np.float64)


def _boxes_of(annos, metric):
    if metric == 0:
        return [np.asarray(a["bbox"], np.float64).reshape(-1, 4) for a in annos]
    out = []
    for a in annos:
        loc, dims, rot = (np.asarray(a[k], np.float64) for k in ("location", "dimensions", "rotation_y"))
        loc, dims = loc.reshape(-1, 3), dims.reshape(-1, 3)
        if metric == 1:
            loc, dims = loc[:, [0, 2]], dims[:, [0, 2]]
        out.append(np.concatenate([loc, dims, rot.reshape(-1, 1)], axis=1))
    return out


def calculate_overlaps(gt_annos, dt_annos, metric, rotate_iou=None):
    """eval_gpu/eval.py:366-437 per image (the reference batches images into parts for its GPU kernel and slices the diagonal
    blocks back out; the blocks are the same).  -> list of [D_i, G_i] (detections x ground truth, the orientation eval_class
    uses: it calls calculate_iou_partly(dt_annos, gt_annos), :511)."""
    fn = {0: lambda b, q: image_box_overlap(b, q), 1: lambda b, q: bev_box_overlap(b, q, -1, rotate_iou),
          2: lambda b, q: d3_box_overlap(b, q, -1, rotate_iou)}.get(metric)
    if fn is None:
        raise ValueError("unknown metric")
    return [fn(d, g) for d, g in zip(_boxes_of(dt_annos, metric), _boxes_of(gt_annos, metric))]


def eval_class(gt_annos, dt_annos, current_classes, difficultys, metric, min_overlaps, compute_aos=False,
               class_names=CLASS_NAMES_FULL, rotate_iou=None):
    """eval_gpu/eval.py:483-599 -> {"recall", "precision", "orientation"}: [class, difficulty, min_overlap, 41]."""
    if len(gt_annos) != len(dt_annos):
        raise ValueError("gt_annos and dt_annos must pair up image by image")
    overlaps = calculate_overlaps(gt_annos, dt_annos, metric, rotate_iou)
    shape = [len(current_classes), len(difficultys), len(min_overlaps), N_SAMPLE_PTS]
    precision, recall, aos = np.zeros(shape), np.zeros(shape), np.zeros(shape)
    for m, cls in enumerate(current_classes):
        for n, difficulty in enumerate(difficultys):
            gt_datas, dt_datas, ign_gts, ign_dts, dcs = [], [], [], [], []
            total_valid = 0
            for g, d in zip(gt_annos, dt_annos):
                nv, ig, idt, dc = clean_data(g, d, cls, difficulty, class_names)
                total_valid += nv
                ign_gts.append(np.array(ig, np.int64))
                ign_dts.append(np.array(idt, np.int64))
                dcs.append(np.stack(dc, 0).astype(np.float64) if dc else np.zeros((0, 4)))
                gt_datas.append(np.concatenate([np.asarray(g["bbox"], np.float64).reshape(-1, 4), np.asarray(g["alpha"], np.float64).reshape(-1, 1)], 1))
                dt_datas.append(np.concatenate([np.asarray(d["bbox"], np.float64).reshape(-1, 4), np.asarray(d["alpha"], np.float64).reshape(-1, 1),
                                                np.asarray(d["score"], np.float64).reshape(-1, 1)], 1))
            for k, min_overlap in enumerate(min_overlaps[:, metric, m]):
                scores = [compute_statistics(overlaps[i], gt_datas[i], dt_datas[i], ign_gts[i], ign_dts[i], dcs[i], metric, min_overlap, 0.0, False)[4]
                          for i in range(len(gt_annos))]
                thresholds = np.array(get_thresholds(np.concatenate(scores) if scores else np.zeros(0), total_valid))
                pr = np.zeros([len(thresholds), 4])
                for i in range(len(gt_annos)):
                    for t, thresh in enumerate(thresholds):
                        tp, fp, fn, sim, _ = compute_statistics(overlaps[i], gt_datas[i], dt_datas[i], ign_gts[i], ign_dts[i], dcs[i], metric,
                                                                min_overlap, thresh, True, compute_aos)
                        pr[t, 0] += tp
                        pr[t, 1] += fp
                        pr[t, 2] += fn
                        if sim != -1:
                            pr[t, 3] += sim
                nt = len(thresholds)
                with np.errstate(divide="ignore", invalid="ignore"):
                    recall[m, n, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 2])
                    precision[m, n, k, :nt] = pr[:, 0] / (pr[:, 0] + pr[:, 1])
                    if compute_aos:
                        aos[m, n, k, :nt] = pr[:, 3] / (pr[:, 0] + pr[:, 1])
                for i in range(nt):      # the envelope runs over the whole 41-long row (zeros past nt), as in :590-594
                    precision[m, n, k, i] = np.max(precision[m, n, k, i:])
                    recall[m, n, k, i] = np.max(recall[m, n, k, i:])
                    if compute_aos:
                        aos[m, n, k, i] = np.max(aos[m, n, k, i:])
    return {"recall": recall, "precision": precision, "orientation": aos}


def get_map(prec):
    """11-point AP (eval_gpu/eval.py:602-606, eval_utils.py:609-614)."""
    return sum(prec[..., i] for i in range(0, prec.shape[-1], 4)) / 11 * 100


def get_map_r40(prec):
    """40-point AP (eval_gpu/eval.py:609-613)."""
    return sum(prec[..., i] for i in range(1, prec.shape[-1])) / 40 * 100


_MIN_OVERLAPS_FULL = np.stack([np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.7]] * 3),
                               np.array([[0.7, 0.5, 0.5, 0.7, 0.5, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5], [0.5, 0.25, 0.25, 0.5, 0.25, 0.5]])], 0)
_MIN_OVERLAPS_MS = np.array([[[0.7, 0.5, 0.5, 0.7, 0.5, 0.7, 0.7, 0.7]] * 3])
_NAMES_FULL = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting", 5: "Truck"}
_NAMES_MS = {0: "Car", 1: "Pedestrian", 2: "Cyclist", 3: "Van", 4: "Person_sitting", 5: "car", 6: "tractor", 7: "trailer"}


def get_official_eval_result(gt_annos, dt_annos, current_classes, protocol="full", difficultys=(0, 1, 2), rotate_iou=None):
    """protocol "ms": eval_utils.py:645-702 (what pointpillars/eval.py prints: image-bbox AP, 11 points) -> (text, map_bbox
    [class, difficulty, 1]).  protocol "full": eval_gpu/eval.py:697-885 -> (text, dict of the *_R40 entries); AOS is computed
    when the first non-empty detection annotation carries a real alpha (!= -10), :738-744."""
    ms = protocol == "ms"
    names, table = (_NAMES_MS, _MIN_OVERLAPS_MS) if ms else (_NAMES_FULL, _MIN_OVERLAPS_FULL)
    to_class = {v: k for k, v in names.items()}
    if not isinstance(current_classes, (list, tuple)):
        current_classes = [current_classes]
    classes = [to_class[c] if isinstance(c, str) else c for c in current_classes]
    min_overlaps = table[:, :, classes]
    cn = CLASS_NAMES_MS if ms else CLASS_NAMES_FULL
    if ms:
        ret = eval_class(gt_annos, dt_annos, classes, difficultys, 0, min_overlaps, False, cn)
        map_bbox = get_map(ret["precision"])
        text = "        Easy   Mod    Hard\n"
        for j, c in enumerate(classes):
            for i in range(min_overlaps.shape[0]):
                text += f"{names[c]} " + "AP@{:.2f}, {:.2f}, {:.2f}:".format(*min_overlaps[i, :, j]) + "\n"
                text += f"bbox AP: {map_bbox[j, 0, i]: .2f}, {map_bbox[j, 1, i]: .2f}, {map_bbox[j, 2, i]: .2f}\n"
        return text, map_bbox
    compute_aos = False
    for anno in dt_annos:
        if anno["alpha"].shape[0] != 0:
            compute_aos = bool(anno["alpha"][0] != -10)
            break
    res = {}
    for metric, key in ((0, "bbox"), (1, "bev"), (2, "3d")):
        ret = eval_class(gt_annos, dt_annos, classes, (0, 1, 2), metric, min_overlaps, compute_aos and metric == 0, cn, rotate_iou)
        res[key] = (get_map(ret["precision"]), get_map_r40(ret["precision"]))
        if metric == 0 and compute_aos:
            res["aos"] = (get_map(ret["orientation"]), get_map_r40(ret["orientation"]))
    text, out = "", {}
    for j, c in enumerate(classes):
        for i in range(min_overlaps.shape[0]):
            for r40 in (0, 1):
                text += f"{names[c]} " + ("AP_R40" if r40 else "AP") + "@{: .2f}, {: .2f}, {: .2f}:".format(*min_overlaps[i, :, j]) + "\n"
                for key in ("bbox", "bev", "3d"):
                    v = res[key][r40]
                    text += f"{key} AP: {v[j, 0, i]: .4f}, {v[j, 1, i]: .4f}, {v[j, 2, i]: .4f}\n"
                if compute_aos:
                    v = res["aos"][r40]
                    text += f"aos AP: {v[j, 0, i]: .2f}, {v[j, 1, i]: .2f}, {v[j, 2, i]: .2f}\n"
            if i == 0:
                for key, tag in (("3d", "3d"), ("bev", "bev"), ("bbox", "image")) + ((("aos", "aos"),) if compute_aos else ()):
                    v = res[key][1]
                    for d, dn in enumerate(("easy", "moderate", "hard")):
                        out[f"{names[c]}_{tag}/{dn}_R40"] = float(v[j, d, 0])
    return text, out
