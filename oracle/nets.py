"""torch-CPU fp32 ORACLE for the conv graphs and the two-stage pipeline -- TEST INFRASTRUCTURE ONLY.

Conv/BN/ReLU = torch.nn.functional.conv2d on the fp32 parameters held by the product modules in the
REFERENCE layout (weight [Cout,Cin,kh,kw], BN folded exactly as SURVEY 8(c) states:
w' = w*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps)).  Graph structure follows
minddet/models/centernet/src/resnet.py:109-252 (BasicBlock / Bottleneck / ResNet; zero-pad +
MaxPool2d(3,2)).  FPN / RPN / RoI head have no reference counterpart ("parity unpinned"): they
restate the public definitions, mirroring minddet_amd/graphs.py op for op.

`quant=True` rounds weights and every layer output to bf16 (what the device stores), which isolates
kernel arithmetic error from bf16 storage error in the parity tests; `quant=False` is the plain fp32
baseline that bench.py times as cpu_baseline.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import np_ops
import oracle


def _q(t, quant):
    return t.to(torch.bfloat16).float() if quant else t


def fold(m):
    w = m.weight.float()
    if m.bn is not None:
        gamma, beta, mean, var, eps = m.bn
        scale = gamma / torch.sqrt(var + eps)
        w = w * scale.view(-1, 1, 1, 1)
        b = beta - mean * scale
        if m.bias is not None:
            b = b + m.bias * scale
    else:
        b = m.bias.float() if m.bias is not None else torch.zeros(w.shape[0])
    return w, b


def conv_module(m, x, residual=None, quant=False):
    """x NCHW fp32. Mirrors md_conv2d's epilogue order: bias -> (bf16 round) -> +residual -> ReLU -> round."""
    w, b = fold(m)
    y = F.conv2d(x, _q(w, quant), b, stride=m.stride, padding=m.pad)
    if getattr(m, "act", None) == "silu":  # SiLU first, then the shortcut add (md_conv2d relu code 2)
        y = y * torch.sigmoid(y)
    if residual is not None:
        y = _q(y, quant) + residual
    if m.relu:
        y = torch.relu(y)
    return _q(y, quant)


def deform_conv_module(m, x, quant=False):
    """DCNv2 as wrapped by centernet/src/resnet.py:24-106 (published definition; the MindSpore primitive's arithmetic is not
    in the reference: parity unpinned).  Mirrors the device data path: offset conv -> bf16, modulated bilinear columns ->
    bf16, fp32 GEMM + folded-BN bias -> ReLU -> bf16."""
    n, c, h, w = x.shape
    k, s, p = m.k, m.stride, m.pad
    off = _q(F.conv2d(x, _q(m.offset_weight.float(), quant), m.offset_bias.float(), stride=s, padding=p), quant)
    ho, wo = off.shape[2], off.shape[3]
    wf, bf = fold(m)
    wf = _q(wf, quant)
    ys = torch.arange(ho, dtype=torch.float32).view(1, ho, 1) * s - p
    xs = torch.arange(wo, dtype=torch.float32).view(1, 1, wo) * s - p
    out = torch.zeros((n, wf.shape[0], ho, wo))
    xp = x.reshape(n, c, h * w)
    for t in range(k * k):
        ky, kx = t // k, t % k
        py = ys + ky + off[:, 2 * t]
        px = xs + kx + off[:, 2 * t + 1]
        mask = torch.sigmoid(off[:, 2 * k * k + t])
        inside = (py > -1) & (py < h) & (px > -1) & (px < w)
        y0, x0 = torch.floor(py), torch.floor(px)
        ly, lx = py - y0, px - x0
        val = torch.zeros((n, c, ho, wo))
        for dy_, dx_ in ((0, 0), (0, 1), (1, 0), (1, 1)):
            yy, xx = (y0 + dy_).long(), (x0 + dx_).long()
            wgt = (ly if dy_ else 1 - ly) * (lx if dx_ else 1 - lx)
            ok = inside & (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            idx = (yy.clamp(0, h - 1) * w + xx.clamp(0, w - 1)).view(n, 1, ho * wo).expand(n, c, ho * wo)
            g = torch.gather(xp, 2, idx).view(n, c, ho, wo)
            val = val + (wgt * ok).unsqueeze(1) * g
        col = _q(val * mask.unsqueeze(1), quant)
        out = out + torch.einsum("oc,nchw->nohw", wf[:, :, ky, kx], col)
    out = out + bf.view(1, -1, 1, 1)
    if m.relu:
        out = torch.relu(out)
    return _q(out, quant)


def _one_rounding_block(blk):
    """True where the device runs the block's last conv and its downsample conv as ONE GEMM (md_conv1x1_dual: the first block of ResNet
    stages 2-4) and therefore stores neither conv3's nor the downsample conv's output: the bf16-matched oracle (`quant`) rounds the
    sum once there.  (Stage 1's md_bottleneck rounds where the layer-by-layer path rounds.)"""
    d = blk.downsample
    if d is None or not hasattr(blk, "conv3"):
        return False
    c1, c3 = blk.conv1, blk.conv3
    by_md_bottleneck = c1.cout == 64 and c1.cin == 64 and d.stride == 1 and blk.conv2.stride == 1 and c3.cout == 256
    return (c3.k == 1 and c3.stride == 1 and d.k == 1 and d.pad == 0 and c3.cout > 64 and c3.cin % 64 == 0 and d.cin % 64 == 0 and
            not by_md_bottleneck)


def resnet_forward(bb, x, quant=False):
    x = conv_module(bb.conv1, x, quant=quant)
    x = F.max_pool2d(F.pad(x, (1, 1, 1, 1), value=0.0), 3, 2)  # resnet.py:199-204
    outs = []
    for st in bb.stages:
        for blk in st:
            mods = [m for m in blk.modules() if m is not blk.downsample]
            out = x
            for m in mods[:-1]:
                out = conv_module(m, out, quant=quant)
            if quant and _one_rounding_block(blk):
                w3, b3 = fold(mods[-1])
                wd, bd = fold(blk.downsample)
                y = F.conv2d(out, _q(w3, True), b3) + F.conv2d(x, _q(wd, True), bd, stride=blk.downsample.stride)
                x = _q(torch.relu(y) if mods[-1].relu else y, True)
            else:
                res = conv_module(blk.downsample, x, quant=quant) if blk.downsample is not None else x
                x = conv_module(mods[-1], out, residual=res, quant=quant)
        outs.append(x)
    return outs


def fpn_forward(neck, feats, quant=False):
    lats = [conv_module(l, f, quant=quant) for l, f in zip(neck.lateral, feats)]
    for i in range(len(lats) - 1, 0, -1):
        up = F.interpolate(lats[i], size=lats[i - 1].shape[-2:], mode="nearest")
        lats[i - 1] = _q(lats[i - 1] + up, quant)
    outs = [conv_module(o, l, quant=quant) for o, l in zip(neck.output, lats)]
    while len(outs) < neck.num_outs:
        outs.append(F.max_pool2d(outs[-1], 1, 2))
    return outs


def rpn_heads(rpn, feats, quant=False):
    return [conv_module(rpn.out, conv_module(rpn.conv, f, quant=quant), quant=quant) for f in feats]


# ----------------------------------------------------------------------------- post-conv pipeline (numpy)
def rpn_level_select(head_nhwc, A, k):
    """head_nhwc [H,W,Cp] float; logits = channels [0,A) flattened (loc, a). Returns (idx, cnt)."""
    logits = head_nhwc[..., :A].reshape(-1).astype(np.float32)
    v, i = np_ops.topk_desc_stable(logits, k)
    return i, len(i)


def rpn_level_decode(head_nhwc, anchors, idx, A, img_hw):
    h = head_nhwc.reshape(-1, head_nhwc.shape[-1]).astype(np.float32)
    loc, a = idx // A, idx % A
    logit = h[loc, a]
    d = np.stack([h[loc, A + a * 4 + j] for j in range(4)], -1)
    boxes = np_ops.delta2bbox(anchors[idx], d, max_shape=img_hw)
    score = (1.0 / (1.0 + np.exp(-logit.astype(np.float64)))).astype(np.float32)
    return boxes, score


def proposals_from_lists(boxes, scores, counts, nms_thr, post):
    """boxes [L,B,k,4], scores [L,B,k], counts [L,B] (device outputs or oracle's own).  Returns per image
    (keep masks [L,B,k], merged order indices, rois [B*post,5], cnt [B])."""
    L, B, k = scores.shape
    keep = np.zeros((L, B, k), np.uint8)
    for l in range(L):
        for b in range(B):
            c = int(counts[l, b])
            keep[l, b, :c] = oracle.nms_aligned(boxes[l, b, :c], nms_thr, 0.0, 2)
    mboxes = boxes.transpose(1, 0, 2, 3).reshape(B, L * k, 4)
    mscores = np.where(keep.astype(bool), scores, -np.inf).transpose(1, 0, 2).reshape(B, L * k)
    rois = np.zeros((B * post, 5), np.float32)
    cnt = np.zeros(B, np.int32)
    topi = np.zeros((B, post), np.int32)
    for b in range(B):
        n_valid = int(np.isfinite(mscores[b]).sum())
        v, i = np_ops.topk_desc_stable(mscores[b], post)
        m = min(post, n_valid)
        cnt[b] = m
        topi[b, :m] = i[:m]
        rois[b * post:(b + 1) * post, 0] = b
        rois[b * post:b * post + m, 1:] = mboxes[b, i[:m]]
    return keep, topi, rois, cnt


def softmax_np(x):
    x = x.astype(np.float32)
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(-1, keepdims=True)


def rcnn_candidates(cls_reg, roi_cnt, nc, score_thr, post):
    R = cls_reg.shape[0]
    B = len(roi_cnt)
    p = softmax_np(cls_reg[:, :nc + 1])[:, :nc]
    valid = (np.arange(R) % post) < np.repeat(roi_cnt, post)
    cand = np.where((p > np.float32(score_thr)) & valid[:, None], p, -np.inf).astype(np.float32)
    return cand.reshape(B, post * nc)


def rcnn_finish(cand, cls_reg, rois, nc, reg_offset, img_hw, npre, nms_thr, max_det, post):
    """From candidate scores (device's or oracle's) to packed detections."""
    B = cand.shape[0]
    dets = np.zeros((B, max_det, 6), np.float32)
    count = np.zeros(B, np.int32)
    sel = []
    for b in range(B):
        n_valid = int(np.isfinite(cand[b]).sum())
        v, i = np_ops.topk_desc_stable(np.where(np.isfinite(cand[b]), cand[b], -np.inf), npre)
        m = min(npre, n_valid)
        v, i = v[:m], i[:m]
        j, c = i // nc, i % nc
        r = b * post + j
        d = np.stack([cls_reg[r, reg_offset + c * 4 + t] for t in range(4)], -1).astype(np.float32)
        boxes = np_ops.delta2bbox(rois[r, 1:], d, stds=(0.1, 0.1, 0.2, 0.2), max_shape=img_hw) if m else np.zeros((0, 4), np.float32)
        keep = oracle.nms_aligned(boxes, nms_thr, 0.0, 2, groups=c.astype(np.int32)).astype(bool) if m else np.zeros(0, bool)
        kidx = np.nonzero(keep)[0][:max_det]
        n = len(kidx)
        dets[b, :n, :4] = boxes[kidx]
        dets[b, :n, 4] = v[kidx]
        dets[b, :n, 5] = c[kidx]
        count[b] = n
        sel.append(dict(idx=i, scores=v, boxes=boxes, labels=c, keep=keep))
    return dets, count, sel


def faster_rcnn_forward(model, images_nhwc, quant=False):
    """Whole two-stage inference on the CPU. images_nhwc [B,H,W,>=3] float tensor. Returns (dets, count)."""
    x = images_nhwc[..., :3].permute(0, 3, 1, 2).float().contiguous()
    B, _, H, W = x.shape
    feats = fpn_forward(model.neck, resnet_forward(model.backbone, x, quant), quant)
    rpn, roi = model.rpn_head, model.roi_head
    heads = rpn_heads(rpn, feats, quant)
    sizes = [(f.shape[2], f.shape[3]) for f in feats]
    anchors = np_ops.fpn_anchors(sizes, rpn.strides, rpn.scale, rpn.ratios)
    L, k, A = len(feats), rpn.nms_pre, rpn.A
    boxes = np.zeros((L, B, k, 4), np.float32)
    scores = np.full((L, B, k), -np.inf, np.float32)
    counts = np.zeros((L, B), np.int32)
    o = 0
    for l, hd in enumerate(heads):
        n = sizes[l][0] * sizes[l][1] * A
        h = hd.permute(0, 2, 3, 1).numpy()
        for b in range(B):
            idx, c = rpn_level_select(h[b], A, k)
            bx, sc = rpn_level_decode(h[b], anchors[o:o + n], idx, A, (H, W))
            boxes[l, b, :c], scores[l, b, :c], counts[l, b] = bx, sc, c
        o += n
    _, _, rois, cnt = proposals_from_lists(boxes, scores, counts, rpn.nms_thr, rpn.max_per_img)
    post = rpn.max_per_img
    R = rois.shape[0]
    pooled = np.zeros((R, roi.P, roi.P, roi.C), np.float32)
    lv = np_ops.fpn_level(rois[:, 1:])
    fnp = [f.numpy() for f in feats[:len(roi.strides)]]
    lvl_of = np.clip(lv - 2, 0, len(fnp) - 1)
    for b in range(B):
        for lvl in range(len(fnp)):
            sel = np.nonzero((rois[:, 0] == b) & (lvl_of == lvl))[0]
            if len(sel):
                pooled[sel] = np_ops.roi_align_fast(fnp[lvl][b], rois[sel, 1:], roi.P, 1.0 / roi.strides[lvl],
                                                    roi.sampling, True).transpose(0, 2, 3, 1)
    x2 = torch.from_numpy(pooled.reshape(R, -1))
    x2 = _q(x2, quant)
    for m in (roi.fc1, roi.fc2, roi.fc_out):
        w, b = fold(m)
        x2 = x2 @ _q(w, quant).view(w.shape[0], -1).t() + b
        if m.relu:
            x2 = torch.relu(x2)
        x2 = _q(x2, quant)
    cls_reg = x2.numpy()
    cand = rcnn_candidates(cls_reg, cnt, roi.nc, roi.score_thr, post)
    dets, count, _ = rcnn_finish(cand, cls_reg, rois, roi.nc, roi.reg_offset, (H, W), roi.nms_pre, roi.nms_thr,
                                 roi.max_per_img, post)
    return dets, count


# ----------------------------------------------------------------------------- CenterNet / CenterPoint RPN oracles
def deconv_module(m, x, quant=False):
    """Conv2dTranspose + BN(eval) + ReLU on NCHW fp32 (weight layout [Cin,Cout,k,k])."""
    w = m.weight_t.float()
    if m.bn is not None:
        gamma, beta, mean, var, eps = m.bn
        scale = gamma / torch.sqrt(var + eps)
        w = w * scale.view(1, -1, 1, 1)
        b = beta - mean * scale
    else:
        b = torch.zeros(w.shape[1])
    y = F.conv_transpose2d(x, _q(w, quant), b, stride=m.stride, padding=m.pad)
    if m.relu:
        y = torch.relu(y)
    return _q(y, quant)


def mask_head_forward(mh, feats, dets, quant=False):
    """FCNMaskHead on the CPU.  feats: list of NCHW fp32 pyramid levels; dets [B,D,6] numpy (the boundary tensor: feed the
    DEVICE detections so that upstream fp differences cannot move a RoI).  Returns masks [B,D,2P,2P] float32."""
    B, D = dets.shape[0], dets.shape[1]
    rois = np.concatenate([np.repeat(np.arange(B, dtype=np.float32), D)[:, None], dets.reshape(B * D, 6)[:, :4]], 1)
    R = rois.shape[0]
    fnp = [f.numpy() for f in feats[:len(mh.strides)]]
    pooled = np.zeros((R, mh.C, mh.P, mh.P), np.float32)
    lvl_of = np.clip(np_ops.fpn_level(rois[:, 1:]) - 2, 0, len(fnp) - 1)
    for b in range(B):
        for lvl in range(len(fnp)):
            sel = np.nonzero((rois[:, 0] == b) & (lvl_of == lvl))[0]
            if len(sel):
                pooled[sel] = np_ops.roi_align_fast(fnp[lvl][b], rois[sel, 1:], mh.P, 1.0 / mh.strides[lvl], mh.sampling, True)
    x = _q(torch.from_numpy(pooled), quant)
    for m in mh.convs:
        x = conv_module(m, x, quant=quant)
    x = conv_module(mh.logits, deconv_module(mh.upsample, x, quant), quant=quant)     # [R, nc, 2P, 2P]
    d = dets.reshape(B * D, 6)
    out = np.zeros((R, 2 * mh.P, 2 * mh.P), np.float32)
    for r in range(R):
        if d[r, 4] > 0:
            out[r] = torch.sigmoid(x[r, int(d[r, 5])]).numpy()
    return out.reshape(B, D, 2 * mh.P, 2 * mh.P)


def centernet_features(model, x, quant=False):
    """centernet/src/centernet_det.py:162-174 with the UNFUSED heads (three 3x3 + three 1x1 convs)."""
    f = resnet_forward(model.backbone, x, quant)[-1]
    for m in model.neck:
        if hasattr(m, "weight_t"):
            f = deconv_module(m, f, quant)
        elif hasattr(m, "offset_weight"):
            f = deform_conv_module(m, f, quant)
        else:
            f = conv_module(m, f, quant=quant)
    out = {}
    for name, (c1, c2) in model.heads.items():
        out[name] = conv_module(c2, conv_module(c1, f, quant=quant), quant=quant)
    return out


def rpn_neck_forward(neck, x, quant=False):
    ups = []
    for i, blk in enumerate(neck.blocks):
        for m in blk:
            x = conv_module(m, x, quant=quant)
        if i - neck.up_start >= 0:
            d = neck.deblocks[i - neck.up_start]
            ups.append(deconv_module(d, x, quant) if hasattr(d, "weight_t") else conv_module(d, x, quant=quant))
    return torch.cat(ups, 1)


# ----------------------------------------------------------------------------- YOLOv5 oracle (parity unpinned)
def _c3(blk, x, quant):
    y = conv_module(blk.cv1, x, quant=quant)
    for a, b in blk.m:
        t = conv_module(a, y, quant=quant)
        y = conv_module(b, t, residual=y if blk.shortcut else None, quant=quant)
    return conv_module(blk.cv3, torch.cat([y, conv_module(blk.cv2, x, quant=quant)], 1), quant=quant)


def _sppf(blk, x, quant):
    y = conv_module(blk.cv1, x, quant=quant)
    ys = [y]
    for _ in range(3):
        ys.append(F.max_pool2d(ys[-1], blk.k, 1, blk.k // 2))
    return conv_module(blk.cv2, torch.cat(ys, 1), quant=quant)


def yolov5_heads(m, x, quant=False):
    cm = lambda mod, t: conv_module(mod, t, quant=quant)
    x = _c3(m.b2, cm(m.b1, cm(m.b0, x)), quant)
    p3 = _c3(m.b4, cm(m.b3, x), quant)
    p4 = _c3(m.b6, cm(m.b5, p3), quant)
    x = _sppf(m.b9, _c3(m.b8, cm(m.b7, p4), quant), quant)
    h10 = cm(m.h10, x)
    h14 = cm(m.h14, _c3(m.h13, torch.cat([F.interpolate(h10, scale_factor=2, mode="nearest"), p4], 1), quant))
    o3 = _c3(m.h17, torch.cat([F.interpolate(h14, scale_factor=2, mode="nearest"), p3], 1), quant)
    o4 = _c3(m.h20, torch.cat([cm(m.h18, o3), h14], 1), quant)
    o5 = _c3(m.h23, torch.cat([cm(m.h21, o4), h10], 1), quant)
    return [cm(d, o) for d, o in zip(m.detect, (o3, o4, o5))]


def yolo_decode_np(head_nhwc, nc, na, stride, anchors, conf_thres):
    """head [B,H,W,C>=na*(5+nc)] float32 -> boxes [B,HWA,4], scores [B,HWA] (-inf where rejected), labels."""
    B, H, W, _ = head_nhwc.shape
    h = head_nhwc[..., :na * (5 + nc)].reshape(B, H, W, na, 5 + nc).astype(np.float32)
    s = (1.0 / (1.0 + np.exp(-h.astype(np.float64)))).astype(np.float32)
    gy, gx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    cx = (s[..., 0] * 2 - 0.5 + gx[None, :, :, None]) * np.float32(stride)
    cy = (s[..., 1] * 2 - 0.5 + gy[None, :, :, None]) * np.float32(stride)
    an = np.asarray(anchors, np.float32).reshape(na, 2)
    w = (s[..., 2] * 2) ** 2 * an[None, None, None, :, 0]
    hh = (s[..., 3] * 2) ** 2 * an[None, None, None, :, 1]
    obj, cls = s[..., 4], s[..., 5:]
    best, lab = cls.max(-1), cls.argmax(-1)
    conf = obj * best
    ok = (obj > np.float32(conf_thres)) & (conf > np.float32(conf_thres))
    boxes = np.stack([cx - w / 2, cy - hh / 2, cx + w / 2, cy + hh / 2], -1).reshape(B, -1, 4).astype(np.float32)
    return boxes, np.where(ok, conf, -np.inf).reshape(B, -1).astype(np.float32), lab.reshape(B, -1).astype(np.int32)


# ----------------------------------------------------------------------------- YOLOv8 oracle (parity unpinned)
def _c2f(blk, x, quant):
    ys = [conv_module(blk.cv1a, x, quant=quant), conv_module(blk.cv1b, x, quant=quant)]
    for a, b in blk.m:
        t = conv_module(a, ys[-1], quant=quant)
        ys.append(conv_module(b, t, residual=ys[-1] if blk.shortcut else None, quant=quant))
    return conv_module(blk.cv2, torch.cat(ys, 1), quant=quant)


def yolov8_heads(m, x, quant=False):
    """-> per level NCHW [B, 4*reg_max + nc, H, W] (box distribution logits, then class logits)."""
    cm = lambda mod, t: conv_module(mod, t, quant=quant)
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    x = _c2f(m.b2, cm(m.b1, cm(m.b0, x)), quant)
    p3 = _c2f(m.b4, cm(m.b3, x), quant)
    p4 = _c2f(m.b6, cm(m.b5, p3), quant)
    p5 = _sppf(m.b9, _c2f(m.b8, cm(m.b7, p4), quant), quant)
    h12 = _c2f(m.h12, torch.cat([up(p5), p4], 1), quant)
    o3 = _c2f(m.h15, torch.cat([up(h12), p3], 1), quant)
    o4 = _c2f(m.h18, torch.cat([cm(m.h16, o3), h12], 1), quant)
    o5 = _c2f(m.h21, torch.cat([cm(m.h19, o4), p5], 1), quant)
    outs = []
    for f, bx, cl in zip((o3, o4, o5), m.box, m.cls):
        outs.append(torch.cat([cm(bx[2], cm(bx[1], cm(bx[0], f))), cm(cl[2], cm(cl[1], cm(cl[0], f)))], 1))
    return outs


def yolov8_decode_np(head_nhwc, nc, reg_max, stride, conf_thres):
    """head [B,H,W,C >= 4*reg_max + nc] float32 -> boxes [B,HW,4], scores [B,HW] (-inf where rejected), labels [B,HW]."""
    B, H, W, _ = head_nhwc.shape
    d = head_nhwc[..., :4 * reg_max].reshape(B, H, W, 4, reg_max).astype(np.float32)
    p = np.exp(d - d.max(-1, keepdims=True))
    dist = (p * np.arange(reg_max, dtype=np.float32)).sum(-1) / p.sum(-1)
    gy, gx = np.meshgrid(np.arange(H, dtype=np.float32) + 0.5, np.arange(W, dtype=np.float32) + 0.5, indexing="ij")
    s = np.float32(stride)
    boxes = np.stack([(gx[None] - dist[..., 0]) * s, (gy[None] - dist[..., 1]) * s, (gx[None] + dist[..., 2]) * s,
                      (gy[None] + dist[..., 3]) * s], -1).reshape(B, -1, 4).astype(np.float32)
    cls = head_nhwc[..., 4 * reg_max:4 * reg_max + nc].astype(np.float32)
    lab = cls.argmax(-1)
    conf = (1.0 / (1.0 + np.exp(-cls.max(-1).astype(np.float64)))).astype(np.float32)
    return boxes, np.where(conf > np.float32(conf_thres), conf, -np.inf).reshape(B, -1).astype(np.float32), lab.reshape(B, -1).astype(np.int32)
