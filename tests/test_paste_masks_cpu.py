"""not gpu: the mask-pasting oracle (oracle/np_ops.py::paste_masks, the checker of md_paste_masks) against the PUBLIC definition it
restates -- mmdet _do_paste_mask / torchvision paste_masks_in_image: F.grid_sample(align_corners=False, zero padding) at pixel centres,
then >= threshold.  The reference has no Mask R-CNN code (README bullet): parity unpinned; this pins the oracle to the public op."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import np_ops


def _public(masks, dets, H, W, thr):
    out = np.zeros((masks.shape[0], H, W), np.uint8)
    for r in range(masks.shape[0]):
        x0, y0, x1, y1 = (float(v) for v in dets[r, :4])
        if not (dets[r, 4] > 0 and x1 > x0 and y1 > y0):
            continue
        iy = (torch.arange(H, dtype=torch.float32) + 0.5 - y0) / (y1 - y0) * 2 - 1
        ix = (torch.arange(W, dtype=torch.float32) + 0.5 - x0) / (x1 - x0) * 2 - 1
        g = torch.stack([ix[None, :].expand(H, W), iy[:, None].expand(H, W)], -1)[None]
        t = F.grid_sample(torch.from_numpy(masks[r])[None, None], g, align_corners=False)[0, 0]
        out[r] = (t >= thr).numpy()
    return out


def test_oracle_equals_grid_sample_definition():
    rng = np.random.default_rng(3)
    H, W, S = 75, 101, 28
    R = 24
    masks = rng.uniform(0, 1, (R, S, S)).astype(np.float32)
    masks[1] = 1.0                                             # a full mask: the pasted shape is the box's pixel support
    dets = np.zeros((R, 6), np.float32)
    cx, cy = rng.uniform(0, W, R), rng.uniform(0, H, R)
    bw, bh = np.exp(rng.uniform(np.log(2), np.log(120), R)), np.exp(rng.uniform(np.log(2), np.log(120), R))
    dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3] = cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2   # partly outside the image
    dets[:, 4] = rng.uniform(0.05, 1, R)
    dets[:, 5] = rng.integers(0, 80, R)
    dets[2, 4] = 0.0                                           # an empty detection slot
    dets[3, 2] = dets[3, 0]                                    # a degenerate box
    dets[4, :4] = (0.0, 0.0, W, H)                             # the whole image
    got = np_ops.paste_masks(masks, dets, (H, W), 0.5)
    ref = _public(masks, dets, H, W, 0.5)
    # the two differ only where the interpolated value sits within fp32 rounding of the threshold
    assert (got != ref).mean() < 1e-5, (got != ref).sum()
    assert got[2].sum() == 0 and got[3].sum() == 0
    ys, xs = np.nonzero(got[1])
    assert ys.min() >= np.floor(dets[1, 1] - 0.5) and xs.max() <= np.ceil(dets[1, 2] + 0.5)
    assert got[4].sum() > 0.3 * H * W
