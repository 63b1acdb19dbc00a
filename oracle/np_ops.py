"""numpy CPU ORACLE for the host-side detection ops -- TEST INFRASTRUCTURE ONLY.

Every function cites the reference text it restates (paths relative to
/root/reference/minddet/models; PP = pointpillars, CN = centernet, CP = centerpoint).
Functions marked "parity unpinned" have no counterpart in the reference; they restate the
public definition of the op (torchvision / mmdet conventions) and are pinned only by their
own known-answer tests.
"""
import math

import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------- box helpers
def limit_period(val, offset=0.5, period=np.pi):
    """PP/src/core/box_np_ops.py:390-392."""
    return val - np.floor(val / period + offset) * period


def center_to_minmax_2d(centers, dims):
    """PP/src/core/box_np_ops.py:377-386 with origin 0.5."""
    return np.concatenate([centers - dims / 2, centers + dims / 2], axis=-1)


def rbbox2d_to_near_bbox(rbboxes):
    """PP/src/core/box_np_ops.py:180-192: rotated (x,y,dx,dy,r) -> nearest axis-aligned box."""
    rots = rbboxes[..., -1]
    rots_0_pi_div_2 = np.abs(limit_period(rots, 0.5, np.pi))
    cond = (rots_0_pi_div_2 > np.pi / 4)[..., np.newaxis]
    bboxes_center = np.where(cond, rbboxes[:, [0, 1, 3, 2]], rbboxes[:, :4])
    return center_to_minmax_2d(bboxes_center[:, :2], bboxes_center[:, 2:])


def center_to_corner_box2d(centers, dims, angles=None):
    """PP/src/core/box_np_ops.py:125-153,258-271,316-341 (origin 0.5).

    corners order (x0y0, x0y1, x1y1, x1y0); rotation matrix [[c,-s],[s,c]] applied as
    points @ R (clockwise for positive angle)."""
    norm = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], dtype=dims.dtype) - np.array(0.5, dims.dtype)
    corners = dims.reshape(-1, 1, 2) * norm.reshape(1, 4, 2)
    if angles is not None:
        s, c = np.sin(angles), np.cos(angles)
        rot = np.stack([[c, -s], [s, c]])
        corners = np.einsum("aij,jka->aik", corners, rot)
    return corners + centers.reshape(-1, 1, 2)


def corner_to_standup_nd(corners):
    """PP/src/core/box_np_ops.py:172-177."""
    return np.concatenate([corners.min(axis=1), corners.max(axis=1)], -1)


# ----------------------------------------------------------------------------- IoU
def iou_jit(boxes, query, eps=0.0):
    """PP/src/core/box_np_ops.py:639-679, vectorised; same float32 op order per element."""
    boxes = np.asarray(boxes)
    query = np.asarray(query)
    dt = boxes.dtype
    e = dt.type(eps)
    qa = (query[:, 2] - query[:, 0] + e) * (query[:, 3] - query[:, 1] + e)
    iw = np.minimum(boxes[:, None, 2], query[None, :, 2]) - np.maximum(boxes[:, None, 0], query[None, :, 0]) + e
    ih = np.minimum(boxes[:, None, 3], query[None, :, 3]) - np.maximum(boxes[:, None, 1], query[None, :, 1]) + e
    ba = (boxes[:, 2] - boxes[:, 0] + e) * (boxes[:, 3] - boxes[:, 1] + e)
    inter = iw * ih
    ua = ba[:, None] + qa[None, :] - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        out = inter / ua
    out = np.where((iw > 0) & (ih > 0), out, dt.type(0))
    return out.astype(dt)


# ----------------------------------------------------------------------------- NMS (python-loop forms)
def nms_jit(dets, thresh, eps=0.0):
    """PP/src/core/nms.py:85-112; dets [N,5] = x1,y1,x2,y2,score (float32). Returns keep list
    of ORIGINAL indices.  argsort()[::-1] tie order is numpy's, as in the reference."""
    x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
    e = dets.dtype.type(eps)
    areas = (x2 - x1 + e) * (y2 - y1 + e)
    order = scores.argsort()[::-1].astype(np.int32)
    n = dets.shape[0]
    sup = np.zeros(n, np.int32)
    keep = []
    zero = dets.dtype.type(0)
    for _i in range(n):
        i = order[_i]
        if sup[i] == 1:
            continue
        keep.append(int(i))
        for _j in range(_i + 1, n):
            j = order[_j]
            if sup[j] == 1:
                continue
            w = max(min(x2[i], x2[j]) - max(x1[i], x1[j]) + e, zero)
            h = max(min(y2[i], y2[j]) - max(y1[i], y1[j]) + e, zero)
            inter = w * h
            ovr = inter / (areas[i] + areas[j] - inter)
            if ovr >= thresh:
                sup[j] = 1
    return keep


def apply_nms(all_boxes, all_scores, thres, max_boxes):
    """PP/src/core/nms.py:7-41 minus the .asnumpy() lines; boxes are (y1,x1,y2,x2)."""
    y1, x1, y2, x2 = (all_boxes[:, i] for i in range(4))
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = all_scores.argsort()[::-1]
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        if len(keep) >= max_boxes:
            break
        xx1 = np.maximum(x1[i], x1[order[1:]])
        yy1 = np.maximum(y1[i], y1[order[1:]])
        xx2 = np.minimum(x2[i], x2[order[1:]])
        yy2 = np.minimum(y2[i], y2[order[1:]])
        w = np.maximum(0.0, xx2 - xx1 + 1)
        h = np.maximum(0.0, yy2 - yy1 + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[order[1:]] - inter)
        inds = np.where(ovr <= thres)[0]
        order = order[inds + 1]
    return np.array(keep)


def soft_nms(dets, sigma=0.5, Nt=0.5, threshold=0.001, method=2):
    """Soft-NMS (Bodla et al. 2017) as published in xingyizhou/CenterNet
    src/lib/external/nms.pyx `soft_nms` -- NOT vendored in the reference (call site
    CN/src/post_process.py:45-52, method=2 gaussian, Nt=0.5, threshold=0.001).
    parity unpinned.  In-place on dets [N,5] (x1,y1,x2,y2,score); returns kept row count;
    rows [0:count] are the survivors (swapped in place, as the published code does)."""
    N = dets.shape[0]
    i = 0
    while i < N:
        maxscore = dets[i, 4]
        maxpos = i
        pos = i + 1
        while pos < N:
            if maxscore < dets[pos, 4]:
                maxscore = dets[pos, 4]
                maxpos = pos
            pos += 1
        if maxpos != i:  # swap whole rows (extra columns, e.g. an id tag, travel with the box)
            tmp = dets[i].copy()
            dets[i] = dets[maxpos]
            dets[maxpos] = tmp
        tx1, ty1, tx2, ty2 = dets[i, :4]
        pos = i + 1
        while pos < N:
            x1, y1, x2, y2 = dets[pos, :4]
            area = (x2 - x1 + 1) * (y2 - y1 + 1)
            iw = min(tx2, x2) - max(tx1, x1) + 1
            if iw > 0:
                ih = min(ty2, y2) - max(ty1, y1) + 1
                if ih > 0:
                    ua = float((tx2 - tx1 + 1) * (ty2 - ty1 + 1) + area - iw * ih)
                    ov = iw * ih / ua
                    if method == 1:
                        weight = 1 - ov if ov > Nt else 1
                    elif method == 2:
                        weight = np.exp(-(ov * ov) / sigma)
                    else:
                        weight = 0 if ov > Nt else 1
                    dets[pos, 4] = weight * dets[pos, 4]
                    if dets[pos, 4] < threshold:
                        dets[pos] = dets[N - 1]
                        N -= 1
                        pos -= 1
            pos += 1
        i += 1
    return N


# ----------------------------------------------------------------------------- anchors
def create_anchors_3d_stride(feature_size, sizes=(1.6, 3.9, 1.56), anchor_strides=(0.4, 0.4, 0.0),
                             anchor_offsets=(0.2, -39.8, -1.78), rotations=(0, np.pi / 2),
                             anchor_range=(0.0, -39.68, -3.0, 69.12, 39.68, 1.0), dtype=np.float32):
    """PP/src/core/box_np_ops.py:453-523.  ms.ops.meshgrid(indexing='ij') restated with
    np.meshgrid.  Note the reference hard-codes num_anchor_size=1, num_anchor_rotation=2
    (:492) -- kept."""
    grid_size = [feature_size[2], feature_size[1]]
    x_stride = (anchor_range[3] - anchor_range[0]) / (grid_size[0] - 1)
    y_stride = (anchor_range[4] - anchor_range[1]) / (grid_size[1] - 1)
    x_shifts = np.arange(anchor_range[0], anchor_range[3] + 1e-5, step=x_stride, dtype=np.float32)
    y_shifts = np.arange(anchor_range[1], anchor_range[4] + 1e-5, step=y_stride, dtype=np.float32)
    z_shifts = np.array([anchor_offsets[2]], dtype=np.float32)
    num_anchor_size, num_anchor_rotation = 1, 2
    anchor_rotation = np.array(rotations, dtype=dtype)
    anchor_size = np.reshape(np.array(sizes, dtype=dtype), [-1, 3])
    xs, ys, zs = np.meshgrid(x_shifts, y_shifts, z_shifts, indexing="ij")
    anchors = np.stack((xs, ys, zs), axis=-1)
    anchors = np.tile(anchors[:, :, :, None, :], (1, 1, 1, anchor_size.shape[0], 1))
    anchor_size = np.tile(anchor_size.reshape(1, 1, 1, -1, 3), ([*anchors.shape[0:3], 1, 1]))
    anchors = np.concatenate((anchors, anchor_size), axis=-1)
    anchors = np.tile(anchors[:, :, :, :, None, :], (1, 1, 1, 1, num_anchor_rotation, 1))
    anchor_rotation = np.tile(anchor_rotation.reshape(1, 1, 1, 1, -1, 1),
                              ([*anchors.shape[0:3], num_anchor_size, 1, 1]))
    anchors = np.concatenate((anchors, anchor_rotation), axis=-1)
    return np.transpose(anchors, [2, 1, 0, 3, 4, 5])


def linspace_f32(lo, hi, n, mode=0):
    """np.linspace(lo, hi, n, dtype=float32) for float32 scalars lo / hi, restated from numpy's source so that BOTH arithmetic
    regimes are available whatever numpy runs the test: mode 0 = float32 products and sums (numpy >= 2, NEP 50), mode 1 = float64
    rounded once (numpy 1.21, the reference's pin: result_type(f32, f32, float(num)) is float64 there).  step is float32 in both:
    f32(f32(hi - lo) / (n - 1)); the last element is hi exactly."""
    lo, hi = np.float32(lo), np.float32(hi)
    i = np.arange(n)
    if n == 1:
        return np.array([lo], np.float32)
    delta = np.float32(hi - lo)
    step = np.float32(delta / np.float32(n - 1))
    if mode == 0:
        f = i.astype(np.float32)
        y = (f / np.float32(n - 1)) * delta if step == 0 else f * step
        y = (y.astype(np.float32) + lo).astype(np.float32)
    else:
        f = i.astype(np.float64)
        y = (f / np.float64(n - 1)) * np.float64(delta) if step == 0 else f * np.float64(step)
        y = (y + np.float64(lo)).astype(np.float32)
    y[-1] = hi
    return y


def create_anchors_3d_range(feature_size, anchor_range, sizes=(1.6, 3.9, 1.56),
                            rotations=(0, np.pi / 2), dtype=np.float32, linspace_mode=0):
    """PP/src/core/box_np_ops.py:526-568: [D,H,W,S,R,7] = (x, y, z centre on np.linspace per axis, size triple, rotation)."""
    anchor_range = np.array(anchor_range, dtype)
    z_centers = linspace_f32(anchor_range[2], anchor_range[5], feature_size[0], linspace_mode)
    y_centers = linspace_f32(anchor_range[1], anchor_range[4], feature_size[1], linspace_mode)
    x_centers = linspace_f32(anchor_range[0], anchor_range[3], feature_size[2], linspace_mode)
    sizes = np.reshape(np.array(sizes, dtype=dtype), [-1, 3])
    rotations = np.array(rotations, dtype=dtype)
    D, H, W, S, R = len(z_centers), len(y_centers), len(x_centers), sizes.shape[0], len(rotations)
    out = np.zeros((D, H, W, S, R, 7), dtype)
    out[..., 0] = x_centers[None, None, :, None, None]
    out[..., 1] = y_centers[None, :, None, None, None]
    out[..., 2] = z_centers[:, None, None, None, None]
    out[..., 3:6] = sizes[None, None, None, :, None, :]
    out[..., 6] = rotations[None, None, None, None, :]
    return out


def generate_anchors(generators, feature_map_size):
    """TargetAssigner.generate_anchors (PP/src/core/target_assigner.py:227-249): generators = list of dicts of
    create_anchors_3d_stride keyword arguments + match_threshold / unmatch_threshold."""
    tabs, match, unmatch = [], [], []
    for g in generators:
        kw = {k: v for k, v in g.items() if k not in ("match_threshold", "unmatch_threshold")}
        a = create_anchors_3d_stride(feature_map_size, **kw)
        a = a.reshape([*a.shape[:3], -1, 7])
        tabs.append(a)
        n = int(np.prod(a.shape[:-1]))
        match.append(np.full([n], g["match_threshold"], a.dtype))
        unmatch.append(np.full([n], g["unmatch_threshold"], a.dtype))
    return {"anchors": np.concatenate(tabs, axis=-2), "matched_thresholds": np.concatenate(match), "unmatched_thresholds": np.concatenate(unmatch)}


def anchors_mask(coors, grid_size_xy, anchors_bv, voxel_size, pc_range, area_threshold):
    """PP/src/data/preprocess.py:211-225 + PP/src/core/box_np_ops.py:745-776.

    coors [V,3] int (z,y,x); grid_size_xy = (nx, ny); returns (anchors_area f32, mask bool)."""
    nx, ny = int(grid_size_xy[0]), int(grid_size_xy[1])
    dense = np.zeros((ny, nx), np.float32)
    np.add.at(dense, (coors[:, 1], coors[:, 2]), np.float32(1))
    dense = dense.cumsum(0)
    dense = dense.cumsum(1)
    stride = np.asarray(voxel_size)
    offset = np.asarray(pc_range)
    c0 = np.floor((anchors_bv[:, 0] - offset[0]) / stride[0]).astype(np.int32)
    c1 = np.floor((anchors_bv[:, 1] - offset[1]) / stride[1]).astype(np.int32)
    c2 = np.floor((anchors_bv[:, 2] - offset[0]) / stride[0]).astype(np.int32)
    c3 = np.floor((anchors_bv[:, 3] - offset[1]) / stride[1]).astype(np.int32)
    c0 = np.maximum(c0, 0)
    c1 = np.maximum(c1, 0)
    c2 = np.minimum(c2, nx - 1)
    c3 = np.minimum(c3, ny - 1)
    area = dense[c3, c2] - dense[c3, c0] - dense[c1, c2] + dense[c1, c0]
    return area.astype(np.float32), area > area_threshold


def fpn_anchors(feat_sizes, strides=(4, 8, 16, 32, 64), scale=8.0, ratios=(0.5, 1.0, 2.0)):
    """FPN/RPN 2-D anchors -- parity unpinned (absent from the reference; SURVEY a6 dagger).
    mmdet AnchorGenerator convention: base size = stride, centre offset 0, per location the
    ratio index varies fastest among the A anchors; locations row-major (y outer, x inner);
    levels concatenated.  h_ratio = sqrt(r), w_ratio = 1/sqrt(r)."""
    out = []
    for (h, w), s in zip(feat_sizes, strides):
        r = np.asarray(ratios, np.float32)
        hr = np.sqrt(r)
        wr = (np.float32(1) / hr).astype(np.float32)
        ws = (np.float32(s) * wr * np.float32(scale)).astype(np.float32)
        hs = (np.float32(s) * hr * np.float32(scale)).astype(np.float32)
        base = np.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], -1).astype(np.float32)  # [A,4]
        sx = (np.arange(w, dtype=np.float32) * np.float32(s))
        sy = (np.arange(h, dtype=np.float32) * np.float32(s))
        yy, xx = np.meshgrid(sy, sx, indexing="ij")
        shifts = np.stack([xx, yy, xx, yy], -1).reshape(-1, 1, 4)
        out.append((shifts + base[None]).reshape(-1, 4).astype(np.float32))
    return np.concatenate(out, 0)


# ----------------------------------------------------------------------------- box codecs
def second_box_encode(boxes, anchors):
    """PP/src/core/box_np_ops.py:8-37 (default flags)."""
    xa, ya, za, wa, la, ha, ra = np.split(anchors, 7, axis=-1)
    xg, yg, zg, wg, lg, hg, rg = np.split(boxes, 7, axis=-1)
    zg = zg + hg / 2
    za = za + ha / 2
    diagonal = np.sqrt(la**2 + wa**2)
    xt = (xg - xa) / diagonal
    yt = (yg - ya) / diagonal
    zt = (zg - za) / ha
    lt = np.log(lg / la)
    wt = np.log(wg / wa)
    ht = np.log(hg / ha)
    rt = rg - ra
    return np.concatenate([xt, yt, zt, wt, lt, ht, rt], axis=-1)


def second_box_decode(enc, anchors):
    """PP/src/core/box_np_ops.py:40-67 / PP/src/core/box_ops.py:47-85 (default flags)."""
    xa, ya, za, wa, la, ha, ra = np.split(anchors, 7, axis=-1)
    xt, yt, zt, wt, lt, ht, rt = np.split(enc, 7, axis=-1)
    za = za + ha / 2
    diagonal = np.sqrt(la**2 + wa**2)
    xg = xt * diagonal + xa
    yg = yt * diagonal + ya
    zg = zt * ha + za
    lg = np.exp(lt) * la
    wg = np.exp(wt) * wa
    hg = np.exp(ht) * ha
    rg = rt + ra
    zg = zg - hg / 2
    return np.concatenate([xg, yg, zg, wg, lg, hg, rg], axis=-1)


def delta2bbox(rois, deltas, means=(0, 0, 0, 0), stds=(1, 1, 1, 1), max_shape=None,
               wh_ratio_clip=16 / 1000):
    """R-CNN delta decode -- parity unpinned (absent from the reference).  mmdet
    legacy-free convention: w = x2-x1, centre = (x1+x2)/2, dw/dh clamped to
    +-|log(wh_ratio_clip)|, output clipped to [0, max_shape]."""
    rois = rois.astype(np.float32)
    d = deltas.astype(np.float32) * np.asarray(stds, np.float32) + np.asarray(means, np.float32)
    mr = np.float32(abs(math.log(wh_ratio_clip)))
    dw = np.clip(d[:, 2], -mr, mr)
    dh = np.clip(d[:, 3], -mr, mr)
    px = (rois[:, 0] + rois[:, 2]) * np.float32(0.5)
    py = (rois[:, 1] + rois[:, 3]) * np.float32(0.5)
    pw = rois[:, 2] - rois[:, 0]
    ph = rois[:, 3] - rois[:, 1]
    gw = pw * np.exp(dw)
    gh = ph * np.exp(dh)
    gx = px + pw * d[:, 0]
    gy = py + ph * d[:, 1]
    x1 = gx - gw * np.float32(0.5)
    y1 = gy - gh * np.float32(0.5)
    x2 = gx + gw * np.float32(0.5)
    y2 = gy + gh * np.float32(0.5)
    if max_shape is not None:
        x1 = np.clip(x1, 0, max_shape[1])
        x2 = np.clip(x2, 0, max_shape[1])
        y1 = np.clip(y1, 0, max_shape[0])
        y2 = np.clip(y2, 0, max_shape[0])
    return np.stack([x1, y1, x2, y2], -1).astype(np.float32)


# ----------------------------------------------------------------------------- CenterNet decode
def centernet_decode(hm, wh, reg, K=100):
    """CN/src/decode.py:40-64 (max-pool NMS), :90-109 (GatherTopK), :151-196 (DetectionDecode).

    hm [B,C,H,W] float32 already sigmoid+clipped (CN/src/utils.py:132-157); wh, reg [B,2,H,W].
    TopK = stable descending sort (ties -> lower index first), SURVEY 8(c).
    Returns detections [B,K,6] = x1,y1,x2,y2,score,cls and the int indices (inds, cls)."""
    B, C, H, W = hm.shape
    pad = np.full((B, C, H + 2, W + 2), -np.inf, hm.dtype)
    pad[:, :, 1:-1, 1:-1] = hm
    hmax = hm.copy()
    for dy in range(3):
        for dx in range(3):
            hmax = np.maximum(hmax, pad[:, :, dy:dy + H, dx:dx + W])
    heat = hm * (hm == hmax).astype(hm.dtype)
    flat = heat.reshape(B, C, H * W)
    order = np.argsort(-flat, axis=2, kind="stable")[:, :, :K]
    tk_scores = np.take_along_axis(flat, order, 2)  # [B,C,K]
    tk_inds = order
    flat2 = tk_scores.reshape(B, C * K)
    order2 = np.argsort(-flat2, axis=1, kind="stable")[:, :K]
    score = np.take_along_axis(flat2, order2, 1)
    cls = order2 // K
    inds = np.take_along_axis(tk_inds.reshape(B, C * K), order2, 1)
    ys = (inds // W).astype(hm.dtype)
    xs = (inds % W).astype(hm.dtype)
    whf = wh.transpose(0, 2, 3, 1).reshape(B, H * W, 2)
    regf = reg.transpose(0, 2, 3, 1).reshape(B, H * W, 2)
    g_wh = np.take_along_axis(whf, inds[:, :, None].repeat(2, 2), 1)
    g_reg = np.take_along_axis(regf, inds[:, :, None].repeat(2, 2), 1)
    xs = xs + g_reg[:, :, 0]
    ys = ys + g_reg[:, :, 1]
    two = hm.dtype.type(2)
    det = np.stack([xs - g_wh[:, :, 0] / two, ys - g_wh[:, :, 1] / two, xs + g_wh[:, :, 0] / two,
                    ys + g_wh[:, :, 1] / two, score, cls.astype(hm.dtype)], -1)
    return det.astype(np.float32), inds.astype(np.int32), cls.astype(np.int32)


def sigmoid_clip(x, lo=1e-4, hi=1 - 1e-4):
    """CN/src/utils.py:132-157."""
    s = (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)
    return np.clip(s, np.float32(lo), np.float32(hi))


def merge_outputs_hard(per_class, max_per_image=100):
    """CN/src/post_process.py:36-61 with SOFT_NMS=False: global score threshold from
    np.partition so that at most max_per_image (ties may exceed) survive."""
    scores = np.hstack([per_class[j][:, 4] for j in sorted(per_class)])
    if len(scores) > max_per_image:
        kth = len(scores) - max_per_image
        thresh = np.partition(scores, kth)[kth]
        per_class = {j: v[v[:, 4] >= thresh] for j, v in per_class.items()}
    return per_class


# ----------------------------------------------------------------------------- top-k select
def topk_desc_stable(scores, k):
    """ops.TopK(sorted=True) stated as a stable descending sort; returns (vals, idx)."""
    idx = np.argsort(-scores, kind="stable")[:k]
    return scores[idx], idx.astype(np.int32)


def pp_select(total_scores, mask, pre_max):
    """PP/src/pointpillars.py:753-765: class max/argmax, masked-out scores := -1, top-k."""
    top_scores = total_scores.max(-1)
    top_labels = total_scores.argmax(-1)
    top_scores = np.where(mask, top_scores, np.float32(-1)).astype(np.float32)
    v, i = topk_desc_stable(top_scores, pre_max)
    return v, i, top_labels.astype(np.int32)


# ----------------------------------------------------------------------------- RoIAlign
def roi_align(feat, rois, out_size, spatial_scale, sampling_ratio=2, aligned=True):
    """RoIAlign -- parity unpinned (absent from the reference; nearest relative is the 4-tap
    bilinear gather CP/det3d_ms/core/utils/center_utils.py:97-131).  torchvision semantics:
    feat [C,H,W] float32, rois [R,4] (x1,y1,x2,y2) in image coords; aligned=True subtracts
    0.5; sample points outside [-1, size] contribute 0; coords clamped to [0, size-1]."""
    C, H, W = feat.shape
    R = rois.shape[0]
    P = out_size
    out = np.zeros((R, C, P, P), np.float32)
    off = 0.5 if aligned else 0.0
    for r in range(R):
        x1, y1, x2, y2 = (np.float32(v) * np.float32(spatial_scale) - np.float32(off) for v in rois[r])
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = max(rw, np.float32(1)), max(rh, np.float32(1))
        bw, bh = rw / np.float32(P), rh / np.float32(P)
        g = sampling_ratio
        for ph in range(P):
            for pw in range(P):
                acc = np.zeros(C, np.float32)
                for iy in range(g):
                    y = y1 + np.float32(ph) * bh + (np.float32(iy) + np.float32(0.5)) * bh / np.float32(g)
                    for ix in range(g):
                        x = x1 + np.float32(pw) * bw + (np.float32(ix) + np.float32(0.5)) * bw / np.float32(g)
                        if y < -1.0 or y > H or x < -1.0 or x > W:
                            continue
                        yy, xx = max(y, np.float32(0)), max(x, np.float32(0))
                        y_lo, x_lo = int(yy), int(xx)
                        if y_lo >= H - 1:
                            y_hi = y_lo = H - 1
                            yy = np.float32(y_lo)
                        else:
                            y_hi = y_lo + 1
                        if x_lo >= W - 1:
                            x_hi = x_lo = W - 1
                            xx = np.float32(x_lo)
                        else:
                            x_hi = x_lo + 1
                        ly, lx = yy - np.float32(y_lo), xx - np.float32(x_lo)
                        hy, hx = np.float32(1) - ly, np.float32(1) - lx
                        acc += (hy * hx) * feat[:, y_lo, x_lo] + (hy * lx) * feat[:, y_lo, x_hi] + \
                               (ly * hx) * feat[:, y_hi, x_lo] + (ly * lx) * feat[:, y_hi, x_hi]
                out[r, :, ph, pw] = acc / np.float32(g * g)
    return out


def fpn_level(rois, k_min=2, k_max=5, canonical=224.0, canonical_level=4):
    """FPN RoI -> level map (Lin et al. 2017, eq. 1): floor(4 + log2(sqrt(wh)/224 + 1e-6)) clamped to [k_min, k_max],
    evaluated WITHOUT a transcendental: level >= k  <=>  sqrt(wh)/224 + 1e-6 >= 2^(k-4)  <=>  wh >= (224*(2^(k-4) - 1e-6))^2.
    The fp32 area is compared with the fp32-rounded thresholds (computed in double), so the integer level is an exact function of
    the fp32 box corners.  parity unpinned (SURVEY a12 dagger: the reference has no FPN)."""
    rois = np.asarray(rois, np.float32)
    w = rois[:, 2] - rois[:, 0]
    h = rois[:, 3] - rois[:, 1]
    area = np.maximum((w * h).astype(np.float32), np.float32(0))
    lvl = np.full(rois.shape[0], k_min, np.int32)
    for k in range(k_min + 1, k_max + 1):
        edge = np.float64(np.float32(canonical)) * (np.ldexp(1.0, k - canonical_level) - 1e-6)
        lvl += (area >= np.float32(edge * edge)).astype(np.int32)
    return lvl


def fpn_level_log2(rois, k_min=2, k_max=5, canonical=224.0, canonical_level=4):
    """The textbook transcendental form (double precision), kept to show the threshold form above is the same map away
    from the level edges."""
    rois = np.asarray(rois, np.float64)
    s = np.sqrt(np.maximum((rois[:, 2] - rois[:, 0]) * (rois[:, 3] - rois[:, 1]), 0.0))
    lvl = np.floor(canonical_level + np.log2(s / canonical + 1e-6))
    return np.clip(lvl, k_min, k_max).astype(np.int32)


def roi_align_fast(feat, rois, out_size, spatial_scale, sampling_ratio=2, aligned=True, chunk=64):
    """Vectorised twin of roi_align (same float32 arithmetic per element, different summation grouping
    is avoided: taps are accumulated in the same (iy, ix) order)."""
    C, H, W = feat.shape
    R, P, g = rois.shape[0], out_size, sampling_ratio
    out = np.zeros((R, C, P, P), np.float32)
    off = np.float32(0.5 if aligned else 0.0)
    f = np.ascontiguousarray(feat.transpose(1, 2, 0))  # [H,W,C]
    for r0 in range(0, R, chunk):
        rr = rois[r0:r0 + chunk].astype(np.float32)
        n = rr.shape[0]
        x1 = rr[:, 0] * np.float32(spatial_scale) - off
        y1 = rr[:, 1] * np.float32(spatial_scale) - off
        x2 = rr[:, 2] * np.float32(spatial_scale) - off
        y2 = rr[:, 3] * np.float32(spatial_scale) - off
        rw, rh = x2 - x1, y2 - y1
        if not aligned:
            rw, rh = np.maximum(rw, np.float32(1)), np.maximum(rh, np.float32(1))
        bw, bh = rw / np.float32(P), rh / np.float32(P)
        pidx = np.arange(P, dtype=np.float32)
        acc = np.zeros((n, P, P, C), np.float32)
        for iy in range(g):
            y = y1[:, None] + pidx[None, :] * bh[:, None] + (np.float32(iy) + np.float32(0.5)) * bh[:, None] / np.float32(g)
            for ix in range(g):
                x = x1[:, None] + pidx[None, :] * bw[:, None] + (np.float32(ix) + np.float32(0.5)) * bw[:, None] / np.float32(g)
                Y = np.broadcast_to(y[:, :, None], (n, P, P))
                X = np.broadcast_to(x[:, None, :], (n, P, P))
                oob = (Y < -1.0) | (Y > H) | (X < -1.0) | (X > W)
                yy, xx = np.maximum(Y, np.float32(0)), np.maximum(X, np.float32(0))
                y_lo, x_lo = yy.astype(np.int32), xx.astype(np.int32)
                ycl, xcl = y_lo >= H - 1, x_lo >= W - 1
                y_lo, x_lo = np.where(ycl, H - 1, y_lo), np.where(xcl, W - 1, x_lo)
                y_hi, x_hi = np.where(ycl, H - 1, y_lo + 1), np.where(xcl, W - 1, x_lo + 1)
                yy = np.where(ycl, y_lo.astype(np.float32), yy)
                xx = np.where(xcl, x_lo.astype(np.float32), xx)
                ly, lx = yy - y_lo.astype(np.float32), xx - x_lo.astype(np.float32)
                hy, hx = np.float32(1) - ly, np.float32(1) - lx
                v = ((hy * hx)[..., None] * f[y_lo, x_lo] + (hy * lx)[..., None] * f[y_lo, x_hi] +
                     (ly * hx)[..., None] * f[y_hi, x_lo] + (ly * lx)[..., None] * f[y_hi, x_hi])
                acc += np.where(oob[..., None], np.float32(0), v)
        out[r0:r0 + n] = (acc / np.float32(g * g)).transpose(0, 3, 1, 2)
    return out


# ----------------------------------------------------------------------------- CenterPoint head
def centerpoint_decode(head, off, ncls, cfg):
    """CP/det3d_ms/models/bbox_heads/center_head.py:297-345 + :398-430 for one task.
    head [B,H,W,C] float32 (attributes at channel offsets `off`).  Returns scores [B,HW] (-1 masked),
    labels [B,HW] (-1), boxes [B,HW,9], nms_boxes [B,HW,7] (dims swapped, heading -rot - pi/2)."""
    B, H, W, _ = head.shape
    f = head.reshape(B, H * W, -1).astype(np.float32)
    hm = (1.0 / (1.0 + np.exp(-f[..., off["hm"]:off["hm"] + ncls]))).astype(np.float32)
    labels = hm.argmax(-1).astype(np.int32)
    scores = hm.max(-1)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    xs = (xs.reshape(1, -1) + f[..., off["reg"]]) * np.float32(cfg["out_size_factor"]) * np.float32(cfg["voxel_size"][0]) + np.float32(cfg["pc_range"][0])
    ys = (ys.reshape(1, -1) + f[..., off["reg"] + 1]) * np.float32(cfg["out_size_factor"]) * np.float32(cfg["voxel_size"][1]) + np.float32(cfg["pc_range"][1])
    zs = f[..., off["height"]]
    dim = np.exp(f[..., off["dim"]:off["dim"] + 3])
    rot = np.arctan2(f[..., off["rot"]], f[..., off["rot"] + 1])
    vel = f[..., off["vel"]:off["vel"] + 2] if off.get("vel", -1) >= 0 else np.zeros((B, H * W, 2), np.float32)
    boxes = np.concatenate([xs[..., None], ys[..., None], zs[..., None], dim, vel, rot[..., None]], -1).astype(np.float32)
    r = np.asarray(cfg["post_center_limit_range"], np.float32)
    mask = (scores > np.float32(cfg["score_threshold"])) & (boxes[..., :3] >= r[:3]).all(-1) & (boxes[..., :3] <= r[3:]).all(-1)
    scores = np.where(mask, scores, np.float32(-1))
    labels = np.where(mask, labels, -1).astype(np.int32)
    boxes = np.where(mask[..., None], boxes, np.float32(0))
    flipped = boxes.copy()
    flipped[..., -1] = -flipped[..., -1] - np.float32(np.pi / 2)
    nms_boxes = flipped[..., [0, 1, 2, 4, 3, 5, -1]]
    return scores, labels, boxes, nms_boxes, mask


def image_preprocess(img_u8, mat, mean, std, out_hw):
    """Restatement of md_image_preprocess: bilinear affine warp (constant-0 border) + (v / 255 - mean) / std, fp32.
    Semantics of cv2.warpAffine(..., flags=INTER_LINEAR) as called at centernet/src/dataset.py:244-247 with the matrix given in
    the output->source direction, but interpolating in floating point (cv2 uses 1/32-pixel fixed point; cv2 is absent
    here: parity unpinned).  img_u8 [N,Hs,Ws,3], mat [N,6] -> [N,Ho,Wo,3] float32."""
    n, hs, ws, _ = img_u8.shape
    ho, wo = out_hw
    out = np.zeros((n, ho, wo, 3), np.float32)
    ys, xs = np.meshgrid(np.arange(ho, dtype=np.float32), np.arange(wo, dtype=np.float32), indexing="ij")
    mean = np.asarray(mean, np.float32)
    std = np.asarray(std, np.float32)
    for b in range(n):
        m = mat[b].astype(np.float32)
        sx = m[0] * xs + m[1] * ys + m[2]
        sy = m[3] * xs + m[4] * ys + m[5]
        x0, y0 = np.floor(sx), np.floor(sy)
        lx, ly = sx - x0, sy - y0
        acc = np.zeros((ho, wo, 3), np.float32)
        for q in range(4):
            yy = (y0 + (q >> 1)).astype(np.int64)
            xx = (x0 + (q & 1)).astype(np.int64)
            w = ((ly if q >> 1 else 1 - ly) * (lx if q & 1 else 1 - lx)).astype(np.float32)
            ok = (yy >= 0) & (yy < hs) & (xx >= 0) & (xx < ws)
            px = img_u8[b, yy.clip(0, hs - 1), xx.clip(0, ws - 1)].astype(np.float32)
            acc += (w * ok)[..., None] * px
        out[b] = (acc * np.float32(1.0 / 255.0) - mean) / std
    return out


# ----------------------------------------------------------------------------- target assignment
def create_target(anchors, gt_boxes, gt_classes, matched_thr, unmatched_thr, anchors_mask=None):
    """Restatement of create_target_np (PP/src/core/target_assigner.py:29-166) as TargetAssigner.assign calls it (:196-224)
    with positive_fraction None: similarity = iou_jit(rbbox2d_to_near_bbox(.), eps 0) (region_similarity.py:46-59), encoding =
    second_box_encode.  matched_thr / unmatched_thr: [A] float32.  Returns labels [A] int32 (-1 ignore / 0 background /
    class), bbox_targets [A,7] float32, bbox_outside_weights [A] float32, gt_ids [A] int32 (-1 where not foreground)."""
    A = anchors.shape[0]
    labels = np.full((A,), -1, np.int32)
    targets = np.zeros((A, 7), np.float32)
    weights = np.zeros((A,), np.float32)
    gt_ids = np.full((A,), -1, np.int32)
    inside = np.arange(A) if anchors_mask is None or len(anchors_mask) == 0 else np.where(anchors_mask)[0]
    a = anchors[inside]
    lab = np.full((len(inside),), -1, np.int32)
    if gt_boxes.shape[0] > 0 and len(inside) > 0:
        ov = iou_jit(rbbox2d_to_near_bbox(a[:, [0, 1, 3, 4, 6]]), rbbox2d_to_near_bbox(gt_boxes[:, [0, 1, 3, 4, 6]]), eps=0.0)
        arg = ov.argmax(axis=1)                      # first maximum, as numpy
        amax = ov[np.arange(len(inside)), arg]
        gmax = ov.max(axis=0).copy()
        gmax[gmax == 0] = -1                         # a ground truth nothing overlaps forces no anchor
        forced = (ov == gmax[None, :]).any(axis=1)
        pos = amax >= matched_thr[inside]
        fg = forced | pos
        lab[fg] = gt_classes[arg[fg]]
        bg = (amax < unmatched_thr[inside]) & ~forced   # labels[bg] = 0 first, then the forced anchors are re-labelled
        lab[bg] = 0
        fgi = np.where(lab > 0)[0]
        t = np.zeros((len(inside), 7), np.float32)
        t[fgi] = second_box_encode(gt_boxes[arg[fgi]], a[fgi])
        targets[inside] = t
        g = np.full((len(inside),), -1, np.int32)
        g[fgi] = arg[fgi]
        gt_ids[inside] = g
    else:
        lab[:] = 0
    labels[inside] = lab
    weights[inside] = (lab > 0).astype(np.float32)
    return labels, targets, weights, gt_ids


def paste_masks(masks, dets, img_hw, threshold=0.5):
    """Mask R-CNN mask pasting (public definition: mmdet _do_paste_mask / torchvision paste_masks_in_image -- grid_sample with
    align_corners=False and zero padding at pixel centres, then mask >= threshold; absent from the reference, whose Mask R-CNN is a README
    bullet: parity unpinned).  Same float32 operation sequence as md_paste_masks (csrc/twostage.hip), every product / sum rounded on its
    own.  masks [R,S,S] f32, dets [R,6] f32 -> [R,H,W] uint8."""
    f32 = np.float32
    masks = np.asarray(masks, f32)
    dets = np.asarray(dets, f32)
    R, S = masks.shape[0], masks.shape[1]
    H, W = img_hw
    out = np.zeros((R, H, W), np.uint8)
    px = np.arange(W, dtype=f32) + f32(0.5)
    py = np.arange(H, dtype=f32) + f32(0.5)
    Sf = f32(S)
    for r in range(R):
        x0, y0 = dets[r, 0], dets[r, 1]
        bw, bh = f32(dets[r, 2] - dets[r, 0]), f32(dets[r, 3] - dets[r, 1])
        if not (dets[r, 4] > 0 and bw > 0 and bh > 0):
            continue
        inv_w, inv_h = f32(1.0) / bw, f32(1.0) / bh
        ix = (((px - x0) * inv_w) * Sf - f32(0.5)).astype(f32)
        iy = (((py - y0) * inv_h) * Sf - f32(0.5)).astype(f32)
        okx, oky = (ix > -1) & (ix < Sf), (iy > -1) & (iy < Sf)
        xl_f, yl_f = np.floor(ix), np.floor(iy)
        fx, fy = (ix - xl_f).astype(f32), (iy - yl_f).astype(f32)
        gx, gy = (f32(1.0) - fx).astype(f32), (f32(1.0) - fy).astype(f32)
        xl, yl = xl_f.astype(np.int64), yl_f.astype(np.int64)
        mp = np.zeros((S + 2, S + 2), f32)      # zero border = the out-of-range taps
        mp[1:-1, 1:-1] = masks[r]
        xi = np.clip(xl + 1, 0, S)              # index of the LEFT tap in the padded mask (clipped where the pixel is outside anyway)
        yi = np.clip(yl + 1, 0, S)
        top = (mp[yi][:, xi] * gx[None, :]).astype(f32) + (mp[yi][:, xi + 1] * fx[None, :]).astype(f32)
        bot = (mp[yi + 1][:, xi] * gx[None, :]).astype(f32) + (mp[yi + 1][:, xi + 1] * fx[None, :]).astype(f32)
        v = (top.astype(f32) * gy[:, None]).astype(f32) + (bot.astype(f32) * fy[:, None]).astype(f32)
        out[r] = ((v.astype(f32) >= f32(threshold)) & oky[:, None] & okx[None, :]).astype(np.uint8)
    return out
