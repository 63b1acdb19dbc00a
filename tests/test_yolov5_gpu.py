"""-m gpu: YOLOv5s graph (build-authored, parity unpinned: YOLO is a README bullet in the reference).
Conv stack (SiLU epilogue, shortcut-after-activation, in-place concat, SPPF, 2x upsample) vs torch fp32 with
bf16 storage; Detect decode vs numpy; top-k / class-aware NMS / packing bit-exact from the device tensors."""
import numpy as np
import pytest
import torch

import oracle
from oracle import nets, np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


def test_yolov5s_end_to_end():
    from minddet.models import Config, build_detector

    cfg = Config.fromfile("configs/yolov5/yolov5s.py")
    cfg.model["conf_thres"] = 0.05  # random-init heads: lower the threshold so that the NMS path is exercised
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    g = torch.Generator().manual_seed(3)
    x = torch.zeros((2, 256, 320, 8))
    x[..., :3] = torch.randn((2, 256, 320, 3), generator=g)
    xb = x.to(torch.bfloat16)
    dets, count, aux = m.forward(xb.to(DEV), return_aux=True)
    torch.cuda.synchronize()
    assert [tuple(h.shape[1:3]) for h in aux["heads"]] == [(32, 40), (16, 20), (8, 10)]
    ref = nets.yolov5_heads(m, xb[..., :3].float().permute(0, 3, 1, 2).contiguous(), quant=True)
    for hd, hr in zip(aux["heads"], ref):
        got = hd.float().cpu().permute(0, 3, 1, 2)[:, :hr.shape[1]]
        rms = hr.pow(2).mean().sqrt().item()
        err = (got - hr).abs().max().item()
        assert err <= 6e-2 * (rms + 0.1 * hr.abs().max().item() + 1e-3), (err, rms)
    # decode from the DEVICE heads
    off = 0
    sd, bd, ld = aux["scores"].cpu().numpy(), aux["boxes"].cpu().numpy(), aux["labels"].cpu().numpy()
    for hd, s, anc in zip(aux["heads"], m.strides, m.ANCHORS):
        b_o, s_o, l_o = nets.yolo_decode_np(hd.float().cpu().numpy(), m.nc, m.na, s, anc, m.conf_thres)
        n = b_o.shape[1]
        s_d = np.where(sd[:, off:off + n] < -1e30, -np.inf, sd[:, off:off + n])
        np.testing.assert_allclose(bd[:, off:off + n], b_o, rtol=2e-5, atol=2e-3)
        both = np.isfinite(s_d) & np.isfinite(s_o)
        assert (np.isfinite(s_d) != np.isfinite(s_o)).mean() < 1e-3
        np.testing.assert_allclose(s_d[both], s_o[both], rtol=3e-6, atol=1e-7)
        assert (ld[:, off:off + n][both] == l_o[both]).mean() > 0.999
        off += n
    # select + class-aware NMS + packing from the DEVICE scores/boxes/labels: exact
    d = dets.cpu().numpy()
    for b in range(2):
        sc = np.where(sd[b] < -1e30, -np.inf, sd[b])
        v, i = np_ops.topk_desc_stable(sc, m.nms_pre)
        k = min(m.nms_pre, int(np.isfinite(sc).sum()))
        v, i = v[:k], i[:k]
        assert int(aux["sel_cnt"][b]) == k
        np.testing.assert_array_equal(aux["sel_idx"].cpu().numpy()[b, :k], i)
        keep = oracle.nms_aligned(bd[b][i], m.iou_thres, 0.0, 2, groups=ld[b][i]).astype(bool)
        kidx = np.nonzero(keep)[0][:m.max_det]
        assert int(count[b]) == len(kidx)
        np.testing.assert_array_equal(d[b, :len(kidx), :4], bd[b][i][kidx])
        np.testing.assert_array_equal(d[b, :len(kidx), 4], v[kidx])
        np.testing.assert_array_equal(d[b, :len(kidx), 5], ld[b][i][kidx])
    assert int(count.sum()) > 0


def test_yolov5s_full_size_batch32_structure():
    """BASELINE.json configs[1] in its stated form: YOLOv5s 640x640 bf16, batch 32 on one GPU -- shapes, padding, score order,
    determinism, select / class-aware NMS / packing bit-exact from the device tensors on sampled images, and the pre-NMS prefix
    flags (conf_thres lowered so that random-init heads produce candidates)."""
    from minddet.models import Config, build_detector
    from minddet_amd.data import synthetic_images

    cfg = Config.fromfile("configs/yolov5/yolov5s.py")
    cfg.model["conf_thres"] = 0.05
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    x = synthetic_images(32, 640, 640, device=DEV)
    dets, count, aux = m.forward(x, return_aux=True)
    dets2, count2 = m.forward(x)
    torch.cuda.synchronize()
    assert [tuple(h.shape[:3]) for h in aux["heads"]] == [(32, 80, 80), (32, 40, 40), (32, 20, 20)]
    assert aux["boxes"].shape == (32, 25200, 4) and dets.shape == (32, m.max_det, 6) and count.shape == (32,)
    assert torch.equal(dets, dets2) and torch.equal(count, count2)              # deterministic
    d, c = dets.cpu().numpy(), count.cpu().numpy()
    sd, bd, ld = aux["scores"].cpu().numpy(), aux["boxes"].cpu().numpy(), aux["labels"].cpu().numpy()
    for b in range(32):
        n = c[b]
        assert 0 <= n <= m.max_det and (np.diff(d[b, :n, 4]) <= 0).all() and (d[b, n:] == 0).all()
        assert (d[b, :n, 2] >= d[b, :n, 0]).all() and (d[b, :n, 3] >= d[b, :n, 1]).all()
    for b in (0, 13, 31):
        sc = np.where(sd[b] < -1e30, -np.inf, sd[b])
        v, i = np_ops.topk_desc_stable(sc, m.nms_pre)
        k = min(m.nms_pre, int(np.isfinite(sc).sum()))
        v, i = v[:k], i[:k]
        assert int(aux["sel_cnt"][b]) == k
        np.testing.assert_array_equal(aux["sel_idx"].cpu().numpy()[b, :k], i)
        keep = oracle.nms_aligned(bd[b][i], m.iou_thres, 0.0, 2, groups=ld[b][i]).astype(bool)
        kidx = np.nonzero(keep)[0][:m.max_det]
        assert c[b] == len(kidx)
        np.testing.assert_array_equal(d[b, :len(kidx), :4], bd[b][i][kidx])
        np.testing.assert_array_equal(d[b, :len(kidx), 4], v[kidx])
        np.testing.assert_array_equal(d[b, :len(kidx), 5], ld[b][i][kidx])
    # prefix flags: an image is flagged iff its prefix was full and produced fewer than max_det survivors
    st = m.prefix_status.tensor(32, DEV).cpu().numpy()
    selc = aux["sel_cnt"].cpu().numpy()
    np.testing.assert_array_equal(st & 1, ((selc >= m.nms_pre) & (c < m.max_det)).astype(st.dtype))


def test_silu_residual_conv_and_slice_writes():
    from minddet_amd import nn_ops
    import torch.nn.functional as F

    g = torch.Generator().manual_seed(0)
    w = torch.randn((64, 64, 3, 3), generator=g) * 0.06
    pc = nn_ops.pack_conv(w, stride=1, pad=1, relu="silu").to(DEV)
    x = torch.randn((1, 20, 24, 64), generator=g).to(torch.bfloat16)
    y = nn_ops.conv2d(x.to(DEV), pc, residual=x.to(DEV)).float().cpu()
    wf = w.to(torch.bfloat16).float()
    c = F.conv2d(x.float().permute(0, 3, 1, 2), wf, None, padding=1)
    ref = ((c * torch.sigmoid(c)).to(torch.bfloat16).float() + x.float().permute(0, 3, 1, 2)).permute(0, 2, 3, 1)
    assert ((y - ref).abs() <= 1.5e-2 * ref.abs() + 1.5e-2).all()
    dst = torch.zeros((1, 40, 48, 96), dtype=torch.bfloat16, device=DEV)
    nn_ops.upsample2x(x.to(DEV), dst, 32)
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="nearest").permute(0, 2, 3, 1)
    assert torch.equal(dst[..., 32:].float().cpu(), up) and float(dst[..., :32].abs().sum()) == 0
    # the source as a channel slice of a wider tensor (r04: the top-down upsample reads a feature where its producer wrote it for the bottom-up concat)
    dst3 = torch.zeros((1, 40, 48, 40), dtype=torch.bfloat16, device=DEV)
    nn_ops.upsample2x(x.to(DEV), dst3, 8, src_c0=16, width=24)
    assert torch.equal(dst3[..., 8:32].float().cpu(), up[..., 16:40]) and float(dst3[..., :8].abs().sum()) == 0 and float(dst3[..., 32:].abs().sum()) == 0
    with pytest.raises(nn_ops._lib.MindDetHipError):
        nn_ops.upsample2x(x.to(DEV), dst3, 8, src_c0=48, width=24)   # slice beyond the source's channels
    dst2 = torch.zeros((1, 20, 24, 128), dtype=torch.bfloat16, device=DEV)
    nn_ops.concat_copy(x.to(DEV), dst2, 64)
    assert torch.equal(dst2[..., 64:].cpu(), x) and float(dst2[..., :64].abs().sum()) == 0
