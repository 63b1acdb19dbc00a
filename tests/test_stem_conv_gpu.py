"""-m gpu: md_stem_conv (the 3-channel stride-2 stem conv of the one-stage detectors on the 4-channel stem layout) vs a plain PyTorch
fp32 conv2d of the same op on bf16-rounded operands, vs md_conv2d on the 8-channel layout, and inside the YOLO graphs.

Tolerance (stated, bf16 path): operands rounded to bf16 on both sides, fp32 accumulation, one bf16 rounding of the output (after the
activation): |err| <= 1.2e-2 * |y| + 1.2e-2 * rms(y), as in tests/test_conv_gpu.py.  The two device paths differ only in the
accumulation order (K = (ky, kx, c) here, (tap, ci) there): they agree within one bf16 ulp."""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


@pytest.mark.parametrize("cfg", [
    # k, cout, act, N, H, W
    (6, 32, "silu", 2, 64, 128), (6, 64, "silu", 1, 128, 64), (3, 64, "silu", 2, 64, 128), (3, 32, "relu", 1, 32, 192),
    (6, 32, None, 3, 48, 64), (3, 64, None, 1, 16, 64),
], ids=str)
def test_stem_conv_vs_torch_fp32_and_vs_md_conv2d(cfg):
    from minddet_amd import nn_ops

    k, cout, act, N, H, W = cfg
    g = torch.Generator().manual_seed(k * 100 + cout + H)
    w = torch.randn((cout, 3, k, k), generator=g) * (2.0 / (k * k * 3)) ** 0.5
    bn = (torch.rand((cout,), generator=g) + 0.5, torch.randn((cout,), generator=g) * 0.1,
          torch.randn((cout,), generator=g) * 0.1, torch.rand((cout,), generator=g) + 0.5, 1e-3)
    pad = 2 if k == 6 else 1
    ps = nn_ops.pack_stem_conv(w, bn=bn, act=act)
    assert ps is not None
    ps.to(DEV)
    x = torch.zeros((N, H, W, 8))
    x[..., :3] = torch.randn((N, H, W, 3), generator=g)
    xb = x.to(torch.bfloat16).to(DEV)
    y = nn_ops.stem_conv(nn_ops.to_stem_layout(xb), ps)
    torch.cuda.synchronize()
    assert tuple(y.shape) == (N, H // 2, W // 2, cout)
    # fp32 reference on the same bf16-rounded operands (the packed, BN-folded weights)
    kx, x0 = (8, 1) if k == 6 else (4, 0)
    wf = ps.w.float().cpu().reshape(cout, k, kx, 4)[:, :, x0:x0 + k, :3].permute(0, 3, 1, 2)
    ref = F.conv2d(xb.float().cpu()[..., :3].permute(0, 3, 1, 2), wf, ps.bias.float().cpu(), stride=2, padding=pad).permute(0, 2, 3, 1)
    ref = F.silu(ref) if act == "silu" else (torch.relu(ref) if act == "relu" else ref)
    got = y.float().cpu()
    rms = ref.pow(2).mean().sqrt().item()
    err = (got - ref).abs()
    assert (err <= 1.2e-2 * ref.abs() + 1.2e-2 * rms).all(), f"max err {err.max().item()} rms {rms}"
    # the generic path on the 8-channel layout
    pc = nn_ops.pack_conv(w, bn=bn, stride=2, pad=pad, relu=act).to(DEV)
    y8 = nn_ops.conv2d(xb, pc).float().cpu()
    assert ((got - y8).abs() <= 8e-3 * y8.abs() + 8e-3 * rms).all()


def test_stem_conv_argument_checks():
    from minddet_amd import _lib, nn_ops

    assert nn_ops.pack_stem_conv(torch.randn((48, 3, 6, 6))) is None      # 48 output channels
    assert nn_ops.pack_stem_conv(torch.randn((32, 3, 5, 5))) is None      # 5x5
    assert nn_ops.pack_stem_conv(torch.randn((32, 4, 3, 3))) is None      # 4 input channels
    ps = nn_ops.pack_stem_conv(torch.randn((32, 3, 6, 6)) * 0.1, act="silu").to(DEV)
    with pytest.raises(_lib.MindDetHipError):   # H not a multiple of 16
        nn_ops.stem_conv(torch.zeros((1, 24 + 16, 64 + 16, 4), dtype=torch.bfloat16, device=DEV), ps)
    with pytest.raises(_lib.MindDetHipError):   # not the stem layout
        nn_ops.stem_conv(torch.zeros((1, 32 + 16, 64 + 16, 8), dtype=torch.bfloat16, device=DEV), ps)
    with pytest.raises(_lib.MindDetHipError):   # fp32 input
        nn_ops.stem_conv(torch.zeros((1, 32 + 16, 64 + 16, 4), dtype=torch.float32, device=DEV), ps)
    ps.kh = 5
    with pytest.raises(_lib.MindDetHipError):
        nn_ops.stem_conv(torch.zeros((1, 32 + 16, 64 + 16, 4), dtype=torch.bfloat16, device=DEV), ps)
    ps.kh = 6
    y = nn_ops.stem_conv(torch.zeros((0, 32 + 16, 64 + 16, 4), dtype=torch.bfloat16, device=DEV), ps)   # empty batch
    assert tuple(y.shape) == (0, 16, 32, 32)


@pytest.mark.parametrize("cfg_path", ["configs/yolov5/yolov5s.py", "configs/yolov8/yolov8l.py"])
def test_yolo_graphs_take_the_stem_layout(cfg_path):
    """The same batch in the 8-channel layout (md_conv2d stem) and in the stem layout (md_stem_conv): head tensors agree within the
    bf16 noise a different accumulation order of the first layer leaves; the stem-layout path is deterministic."""
    from minddet.models import Config, build_detector
    from minddet_amd import nn_ops

    cfg = Config.fromfile(cfg_path)
    m = build_detector(cfg.model, cfg.train_cfg, cfg.test_cfg).to(DEV)
    assert m.stem is not None
    g = torch.Generator().manual_seed(9)
    x = torch.zeros((2, 128, 192, 8))
    x[..., :3] = torch.randn((2, 128, 192, 3), generator=g)
    xb = x.to(torch.bfloat16).to(DEV)
    x4 = nn_ops.to_stem_layout(xb)
    h8 = m.features(xb)
    h4 = m.features(x4)
    h4b = m.features(x4)
    torch.cuda.synchronize()
    for a8, a4, a4b in zip(h8, h4, h4b):
        a8, a4 = (a8[0] if isinstance(a8, (tuple, list)) else a8), (a4[0] if isinstance(a4, (tuple, list)) else a4)
        a4b = a4b[0] if isinstance(a4b, (tuple, list)) else a4b
        assert torch.equal(a4, a4b)
        d = (a8.float() - a4.float()).abs()
        rms = a8.float().pow(2).mean().sqrt().item()
        assert d.max().item() <= 0.1 * (rms + a8.float().abs().max().item()), (d.max().item(), rms)
        assert d.mean().item() <= 1e-2 * (rms + 1e-3)
    dets, count = m.forward(x4)[:2]
    assert dets.shape[0] == 2 and count.shape[0] == 2
