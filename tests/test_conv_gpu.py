"""-m gpu: implicit-GEMM MFMA conv vs a plain PyTorch fp32 conv2d of the same op (CPU).

Tolerance (stated, bf16 path): inputs and folded weights are rounded to bf16 on BOTH sides, the
kernel accumulates in fp32 and rounds the output once to bf16 (twice when a residual is fused),
so |err| <= 2^-8 * |y| + accumulation-order noise: rtol 1.2e-2, atol 1.2e-2 * rms(y)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"

CASES = [
    # name, N, H, W, Cin, Cout, k, stride, pad, relu, residual, bn
    ("1x1_256_64", 2, 25, 42, 256, 64, 1, 1, 0, True, False, True),
    ("3x3_64_64", 1, 50, 84, 64, 64, 3, 1, 1, True, False, True),
    ("3x3_s2_128", 2, 37, 41, 128, 128, 3, 2, 1, True, False, True),
    ("stem7x7", 1, 96, 160, 3, 64, 7, 2, 3, True, False, True),
    ("ds1x1_s2", 2, 26, 44, 256, 512, 1, 2, 0, False, False, True),
    ("res_1x1_64_256", 1, 50, 84, 64, 256, 1, 1, 0, True, True, True),
    ("rpn_head_15", 1, 13, 21, 256, 15, 1, 1, 0, False, False, False),
    ("fpn3x3_256", 1, 25, 42, 256, 256, 3, 1, 1, False, False, False),
    ("fc_12544_1024", 1, 1, 300, 12544, 1024, 1, 1, 0, True, False, False),
    ("cout_40", 1, 20, 20, 64, 40, 3, 1, 1, True, False, True),
    ("tiny", 1, 3, 5, 8, 8, 3, 1, 1, False, False, False),
    ("halo_ragged", 2, 13, 21, 128, 256, 3, 1, 1, True, True, True),
    ("halo_edge", 1, 8, 16, 64, 128, 3, 1, 1, False, False, False),
    ("halo_big", 1, 100, 168, 256, 256, 3, 1, 1, True, False, False),
    ("pp_1x1_1024_512_res", 2, 50, 84, 1024, 512, 1, 1, 0, True, True, True),
    ("pp_3x3_s2_128_256", 1, 51, 85, 128, 256, 3, 2, 1, True, False, True),
    ("halo64_128_64_res", 2, 19, 37, 128, 64, 3, 1, 1, True, True, True),
    ("halo64_192_192", 1, 24, 33, 192, 192, 3, 1, 1, True, False, False),
    ("c32_3x3", 2, 40, 44, 32, 64, 3, 1, 1, True, False, True),
    ("c32_3x3_s2_res", 1, 41, 37, 32, 32, 3, 2, 1, True, True, True),
    ("c32_1x1", 2, 20, 24, 32, 128, 1, 1, 0, False, False, False),
    ("c32_4x4_16_taps", 1, 18, 22, 32, 40, 4, 2, 1, True, False, False),
    ("c32_5x5_generic", 1, 18, 22, 32, 32, 5, 1, 2, True, False, False),
]


@pytest.mark.parametrize("variant", [0, 1, 2, 11, 15, 16, 20, 22, 27])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_vs_torch_fp32(case, variant):
    from minddet_amd import nn_ops

    name, N, H, W, Cin, Cout, k, stride, pad, relu, use_res, use_bn = case
    g = torch.Generator().manual_seed(hash(name) % 10000)
    x = torch.randn((N, Cin, H, W), generator=g)
    w = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    bias = None if use_bn else torch.randn((Cout,), generator=g) * 0.1
    bn = None
    if use_bn:
        bn = (torch.rand((Cout,), generator=g) + 0.5, torch.randn((Cout,), generator=g) * 0.1,
              torch.randn((Cout,), generator=g) * 0.1, torch.rand((Cout,), generator=g) + 0.5, 1e-5)
    korder = 1 if (variant in (2, 11, 16, 27) and Cin % 64 == 0 and k > 1) else 0
    if variant == 16:
        variant = 15  # the ping-pong kernel on korder-1 weights
    pc = nn_ops.pack_conv(w, bias=bias, bn=bn, stride=stride, pad=pad, relu=relu, korder=korder).to(DEV)
    cin_p = pc.cin
    x_nhwc = torch.zeros((N, H, W, cin_p))
    x_nhwc[..., :Cin] = x.permute(0, 2, 3, 1)
    xb = x_nhwc.to(torch.bfloat16)
    ho, wo = nn_ops.conv_out_hw(H, W, pc)
    res = None
    if use_res:
        res = torch.randn((N, ho, wo, pc.cout), generator=g).to(torch.bfloat16)
    y = nn_ops.conv2d(xb.to(DEV), pc, residual=None if res is None else res.to(DEV), variant=variant)
    torch.cuda.synchronize()
    y = y.float().cpu()
    # fp32 reference on the same bf16-rounded operands
    k_real = k * k * cin_p
    wf = pc.w.float().cpu()[:Cout, :k_real]
    if pc.korder == 1:
        wf = wf.reshape(Cout, cin_p // 64, k, k, 64).permute(0, 2, 3, 1, 4).reshape(Cout, k, k, cin_p)
    wf = wf.reshape(Cout, k, k, cin_p).permute(0, 3, 1, 2)
    ref = F.conv2d(xb.float().permute(0, 3, 1, 2), wf, pc.bias.cpu()[:Cout], stride=stride, padding=pad)
    ref = ref.permute(0, 2, 3, 1)
    if use_res:
        if relu:
            pass
        ref = ref.to(torch.bfloat16).float() + res.float()[..., :Cout]
    if relu:
        ref = torch.relu(ref)
    assert y.shape == (N, ho, wo, pc.cout)
    got = y[..., :Cout]
    rms = ref.pow(2).mean().sqrt().item()
    err = (got - ref).abs()
    tol = 1.2e-2 * ref.abs() + 1.2e-2 * rms
    assert (err <= tol).all(), f"max err {err.max().item()} rms {rms}"
    if pc.cout > Cout:  # padded output channels are exactly bias-free zeros (or relu(0))
        assert (y[..., Cout:] == 0).all()


def test_conv_linearity_property_full_size():
    """Size-independent property at a BASELINE-sized layer (R50 layer1 conv2 at 800x1344, batch 2):
    conv(a*x) == a*conv(x) for a power-of-two a (exact in bf16/fp32), no bias."""
    from minddet_amd import nn_ops

    g = torch.Generator().manual_seed(1)
    w = torch.randn((64, 64, 3, 3), generator=g) * 0.05
    pc = nn_ops.pack_conv(w, stride=1, pad=1, relu=False).to(DEV)
    x = torch.randn((2, 200, 336, 64), generator=g).to(torch.bfloat16).to(DEV)
    y1 = nn_ops.conv2d(x, pc)
    y2 = nn_ops.conv2d((x.float() * 4).to(torch.bfloat16), pc)
    assert torch.equal((y1.float() * 4), y2.float())
    # and a spot check against fp32 on a crop (interior pixels only)
    xc = x[:1, 40:72, 100:132].float().cpu().permute(0, 3, 1, 2)
    wf = pc.w.float().cpu()[:64, :576].reshape(64, 3, 3, 64).permute(0, 3, 1, 2)
    ref = F.conv2d(xc, wf, None, padding=0).permute(0, 2, 3, 1)
    got = y1[:1, 41:71, 101:131].float().cpu()
    assert (got - ref).abs().max() <= 1.2e-2 * ref.abs().max()


def test_conv_rejects_bad_args():
    from minddet_amd import _lib, nn_ops

    w = torch.randn((64, 64, 3, 3))
    pc = nn_ops.pack_conv(w, stride=1, pad=1).to(DEV)
    with pytest.raises(_lib.MindDetHipError):
        nn_ops.conv2d(torch.zeros((1, 8, 8, 32), dtype=torch.bfloat16, device=DEV), pc)
    with pytest.raises(_lib.MindDetHipError):
        nn_ops.conv2d(torch.zeros((1, 8, 8, 64), dtype=torch.float32, device=DEV), pc)


def test_conv_fused_upsampled_residual():
    """FPN top-down add fused into the lateral 1x1 conv: y = conv(x) + nearest_up2(top)."""
    from minddet_amd import nn_ops

    g = torch.Generator().manual_seed(5)
    for (h, w) in [(50, 84), (25, 41)]:
        wt = torch.randn((256, 512, 1, 1), generator=g) * 0.05
        bias = torch.randn((256,), generator=g) * 0.1
        pc = nn_ops.pack_conv(wt, bias=bias).to(DEV)
        x = torch.randn((2, h, w, 512), generator=g).to(torch.bfloat16)
        top = torch.randn((2, (h + 1) // 2, (w + 1) // 2, 256), generator=g).to(torch.bfloat16)
        y = nn_ops.conv2d(x.to(DEV), pc, residual=top.to(DEV), res_upsample=True).float().cpu()
        c = F.conv2d(x.float().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(), bias).permute(0, 2, 3, 1)
        up = top.float().repeat_interleave(2, 1).repeat_interleave(2, 2)[:, :h, :w]
        ref = c.to(torch.bfloat16).float() + up
        assert ((y - ref).abs() <= 1.5e-2 * ref.abs() + 1.5e-2).all()


@pytest.mark.parametrize("shape", [(2, 64, 128), (1, 32, 64), (3, 48, 192)])
def test_stem_pool_vs_torch_fp32(shape):
    """md_stem_pool (conv 7x7/2 + BN + ReLU + zero-pad + maxpool 3x3/2 in one launch) vs fp32 torch on the same
    bf16-rounded operands, and vs the two-launch path (md_conv2d + md_maxpool2d).  Tolerance as above (bf16 output)."""
    from minddet_amd import nn_ops

    n, h, w = shape
    g = torch.Generator().manual_seed(11 + h)
    x = torch.randn((n, h, w, 3), generator=g)
    wt = torch.randn((64, 3, 7, 7), generator=g) * (2.0 / 147) ** 0.5
    bn = (torch.rand((64,), generator=g) + 0.5, torch.randn((64,), generator=g) * 0.1,
          torch.randn((64,), generator=g) * 0.1, torch.rand((64,), generator=g) + 0.5, 1e-5)
    x8 = torch.zeros((n, h, w, 8))
    x8[..., :3] = x
    xb = x8.to(torch.bfloat16).to(DEV)
    ps = nn_ops.pack_stem(wt, bn=bn).to(DEV)
    y = nn_ops.stem_pool(nn_ops.to_stem_layout(xb), ps).float().cpu()
    # fp32 reference on the bf16-rounded packed operands
    wf = ps.w.float().cpu().reshape(64, 7, 8, 4)[:, :, :7, :3].permute(0, 3, 1, 2)
    c = F.conv2d(xb.float().cpu()[..., :3].permute(0, 3, 1, 2), wf, ps.bias.cpu(), stride=2, padding=3)
    c = torch.relu(c).to(torch.bfloat16).float()
    ref = F.max_pool2d(F.pad(c, (1, 1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
    assert y.shape == ref.shape == (n, h // 4, w // 4, 64)
    rms = ref.pow(2).mean().sqrt().item()
    assert ((y - ref).abs() <= 1.2e-2 * ref.abs() + 1.2e-2 * rms).all()
    # the generic two-launch path computes the same function
    pc = nn_ops.pack_conv(wt, bn=bn, stride=2, pad=3, relu=True).to(DEV)
    y2 = nn_ops.maxpool2d(nn_ops.conv2d(xb, pc), 3, 2, 1, zero_pad=True).float().cpu()
    assert ((y - y2).abs() <= 1.2e-2 * ref.abs() + 1.2e-2 * rms).all()
    assert (y == y2).float().mean() > 0.98


def test_conv_batch_chunking_is_bit_identical():
    """A batch above the 32-bit DMA reach is run as image chunks; with the limit lowered the chunked result must equal
    the single-launch result bit for bit (residual and all)."""
    import ctypes

    from minddet_amd import _lib, nn_ops

    g = torch.Generator().manual_seed(21)
    wt = torch.randn((128, 64, 3, 3), generator=g) * 0.05
    pc = nn_ops.pack_conv(wt, bias=torch.randn((128,), generator=g) * 0.1, stride=1, pad=1, relu=True).to(DEV)
    x = torch.randn((5, 20, 24, 64), generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn((5, 20, 24, 128), generator=g).to(torch.bfloat16).to(DEV)
    y0 = nn_ops.conv2d(x, pc, residual=r)
    lc = _lib.lib().md_conv2d_launch_count
    lc.restype = ctypes.c_longlong
    l0 = lc()
    y1 = nn_ops.conv2d(x, pc, residual=r, tune=nn_ops.ConvTune(chunk_limit=2 * 20 * 24 * 64 * 2 + 1))  # two images per chunk -> 3 launches
    assert lc() - l0 == 3
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)


@pytest.mark.parametrize("shape", [(2, 96, 96), (1, 13, 21), (3, 75, 91)])
def test_conv2d_head_fused_vs_two_launches(shape):
    """md_conv2d_head (3x3 conv 256 -> 256 + ReLU, then a 1x1 head with 15 (->16) channels): one fused launch where the
    ping-pong kernel applies (first / third shape), two launches through a temporary otherwise (second shape); both must
    agree with md_conv2d twice (same bf16 intermediate, fp32 accumulation in a different order) and with fp32 torch."""
    from minddet_amd import nn_ops

    n, h, w = shape
    g = torch.Generator().manual_seed(31 + h)
    w1 = torch.randn((256, 256, 3, 3), generator=g) * (2.0 / 2304) ** 0.5
    b1 = torch.randn((256,), generator=g) * 0.1
    w2 = torch.randn((15, 256, 1, 1), generator=g) * 0.05
    b2 = torch.randn((15,), generator=g) * 0.1
    pc = nn_ops.pack_conv(w1, bias=b1, stride=1, pad=1, relu=True).to(DEV)
    pc2 = nn_ops.pack_conv(w2, bias=b2).to(DEV)
    x = torch.randn((n, h, w, 256), generator=g).to(torch.bfloat16)
    y = nn_ops.conv2d_head(x.to(DEV), pc, pc2).float().cpu()
    y_two = nn_ops.conv2d(nn_ops.conv2d(x.to(DEV), pc), pc2).float().cpu()
    mid = torch.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w1.to(torch.bfloat16).float(), b1, padding=1)).to(torch.bfloat16).float()
    ref = F.conv2d(mid, w2.to(torch.bfloat16).float(), b2).permute(0, 2, 3, 1)
    assert y.shape == (n, h, w, 16) and (y[..., 15] == 0).all()
    rms = ref.pow(2).mean().sqrt().item()
    assert ((y[..., :15] - ref).abs() <= 1.5e-2 * ref.abs() + 1.5e-2 * rms).all()
    assert ((y - y_two).abs() <= 1.0e-2 * y_two.abs() + 1.0e-2 * rms).all()


def test_conv2d_head_batch_chunking_is_bit_identical():
    """The fused RPN head through the image-chunked path (lowered limit) equals the single-launch result."""
    import ctypes

    from minddet_amd import _lib, nn_ops

    g = torch.Generator().manual_seed(41)
    pc = nn_ops.pack_conv(torch.randn((256, 256, 3, 3), generator=g) * 0.02, bias=torch.randn((256,), generator=g) * 0.1,
                          stride=1, pad=1, relu=True).to(DEV)
    pc2 = nn_ops.pack_conv(torch.randn((15, 256, 1, 1), generator=g) * 0.05, bias=torch.randn((15,), generator=g) * 0.1).to(DEV)
    x = torch.randn((3, 96, 96, 256), generator=g).to(torch.bfloat16).to(DEV)
    y0 = nn_ops.conv2d_head(x, pc, pc2)
    # two images per chunk: the second chunk (one image, 36 tiles) takes the two-launch path
    y1 = nn_ops.conv2d_head(x, pc, pc2, tune=nn_ops.ConvTune(chunk_limit=2 * 96 * 96 * 256 * 2 + 1))
    torch.cuda.synchronize()
    assert torch.equal(y0[:2], y1[:2])
    assert (y0[2].float() - y1[2].float()).abs().max().item() <= 2e-2 * (1 + y0[2].float().abs().max().item())


@pytest.mark.parametrize("cfg", [
    # name, N, H, W, Cin, Cout, k, act, x buffer channels, x offset, residual buffer channels, residual offset, output offset
    ("igemm_1x1_silu", 2, 40, 40, 128, 128, 1, "silu", 320, 64, 256, 128, 128),
    ("halo64_3x3_relu", 2, 33, 47, 64, 64, 3, "relu", 192, 128, 128, 64, 0),
    ("pingpong_3x3_silu", 2, 128, 130, 256, 256, 3, "silu", 512, 256, 768, 512, 256),
    ("igemm_3x3_silu", 1, 37, 53, 128, 128, 3, "silu", 256, 8, 136, 8, 0),
    ("igemm_plain_3x3_relu_x_only", 1, 37, 53, 128, 128, 3, "relu", 256, 128, 0, 0, 0),
    ("generic_k_8ch", 1, 20, 24, 8, 32, 3, "relu", 24, 16, 64, 32, 0),
], ids=lambda c: c[0])
def test_conv_channel_slice_operands_are_bit_identical(cfg):
    """md_conv2d_attrs.x_c_off / res_c_off: a conv reading its input and its residual as channel ranges of wider tensors (the
    split / concat graphs of C2f / C3 without copies) equals the same conv on contiguous copies of the ranges, bit for bit."""
    from minddet_amd import nn_ops

    name, N, H, W, Cin, Cout, k, act, xc, xo, rc, ro, co = cfg
    g = torch.Generator().manual_seed(len(name))
    wt = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    pc = nn_ops.pack_conv(wt, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=k // 2, relu=act).to(DEV)
    xbuf = torch.randn((N, H, W, xc), generator=g).to(torch.bfloat16).to(DEV)
    rbuf = torch.randn((N, H, W, rc), generator=g).to(torch.bfloat16).to(DEV) if rc else None
    x_c = xbuf[..., xo:xo + Cin].contiguous()
    r_c = rbuf[..., ro:ro + Cout].contiguous() if rc else None
    ref = nn_ops.conv2d(x_c, pc, residual=r_c)
    got = nn_ops.conv2d(xbuf, pc, residual=rbuf, x_c_off=xo, res_c_off=ro if rc else None)
    assert torch.equal(got, ref)
    # and straight into a concat buffer
    cat = torch.full((N, H, W, co + Cout + 8), 3.0, dtype=torch.bfloat16, device=DEV)
    nn_ops.conv2d(xbuf, pc, residual=rbuf, x_c_off=xo, res_c_off=ro if rc else None, out=cat, c_off=co)
    assert torch.equal(cat[..., co:co + Cout], ref) and (cat[..., co + Cout:] == 3.0).all() and (cat[..., :co] == 3.0).all()
    # sanity against torch fp32 (the contiguous path itself is covered by test_conv_vs_torch_fp32)
    y = F.conv2d(x_c.float().cpu().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(), pc.bias[:Cout].float().cpu(), padding=k // 2).permute(0, 2, 3, 1)
    y = F.silu(y) if act == "silu" else y
    if rc:
        y = y.to(torch.bfloat16).float() + r_c.float().cpu() if act == "silu" else y + r_c.float().cpu()
    y = torch.relu(y) if act == "relu" else y
    assert ((ref.float().cpu() - y).abs() <= 2e-2 * y.abs() + 2e-2).all()


def test_conv_channel_slice_rejects_bad_ranges():
    from minddet_amd import _lib, nn_ops

    pc = nn_ops.pack_conv(torch.randn((64, 64, 1, 1)) * 0.1).to(DEV)
    x = torch.zeros((1, 8, 8, 128), dtype=torch.bfloat16, device=DEV)
    for off in (4, 72, -8):
        with pytest.raises(_lib.MindDetHipError):
            nn_ops.conv2d(x, pc, x_c_off=off)
    r = torch.zeros((1, 8, 8, 96), dtype=torch.bfloat16, device=DEV)
    for off in (4, 40):
        with pytest.raises(_lib.MindDetHipError):
            nn_ops.conv2d(x, pc, x_c_off=0, residual=r, res_c_off=off)
    with pytest.raises(_lib.MindDetHipError):
        nn_ops.conv2d(x, pc, x_c_off=0, residual=torch.zeros((1, 4, 4, 96), dtype=torch.bfloat16, device=DEV), res_c_off=0)


def test_diagnostic_variants_are_not_in_the_product_library():
    """Variants 17-19 / 25 (timing ablations and stamp builds that do not compute the convolution) exist only in the -DMD_DIAG
    build used by tools/: the product library answers MD_ERR_ARG (rc 2) and leaves the output untouched."""
    from minddet_amd import _lib, nn_ops

    g = torch.Generator().manual_seed(1)
    pc = nn_ops.pack_conv(torch.randn((256, 256, 3, 3), generator=g) * 0.02, pad=1, relu=True, korder=1).to(DEV)
    x = torch.randn((1, 32, 32, 256), generator=g).to(torch.bfloat16).to(DEV)
    out = torch.full((1, 32, 32, 256), 7.0, dtype=torch.bfloat16, device=DEV)
    for v in (17, 18, 19, 25):
        with pytest.raises(_lib.MindDetHipError, match="rc=2"):
            nn_ops.conv2d(x, pc, out=out, variant=v)
    torch.cuda.synchronize()
    assert bool((out == 7.0).all())
    assert not hasattr(_lib.lib(), "md_diag_set_stamp_buffer") and not hasattr(_lib.lib(), "md_conv2d_chain")




BOTTLENECK_CASES = [
    # name, N, H, W, Cin, downsample
    ("identity_whole_tiles", 2, 24, 48, 256, False),
    ("identity_ragged", 1, 19, 23, 256, False),
    ("identity_one_tile", 1, 8, 16, 256, False),
    ("identity_smaller_than_a_tile", 2, 5, 7, 256, False),
    ("first_block_downsample", 2, 24, 40, 64, True),
    ("first_block_ragged", 1, 13, 37, 64, True),
    ("stage1_like", 3, 50, 84, 256, False),
]


def _bottleneck_modules(cin, seed, downsample):
    from minddet_amd import nn_ops

    g = torch.Generator().manual_seed(seed)

    def bn(c):
        return (torch.rand((c,), generator=g) + 0.5, torch.randn((c,), generator=g) * 0.1, torch.randn((c,), generator=g) * 0.1,
                torch.rand((c,), generator=g) + 0.5, 1e-5)

    w1 = torch.randn((64, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
    w2 = torch.randn((64, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5
    w3 = torch.randn((256, 64, 1, 1), generator=g) * (2.0 / 64) ** 0.5
    pcs = [nn_ops.pack_conv(w1, bn=bn(64), relu=True).to(DEV), nn_ops.pack_conv(w2, bn=bn(64), stride=1, pad=1, relu=True).to(DEV),
           nn_ops.pack_conv(w3, bn=bn(256), relu=True).to(DEV)]
    pd = nn_ops.pack_conv(torch.randn((256, cin, 1, 1), generator=g) * (1.0 / cin) ** 0.5, bn=bn(256), relu=False).to(DEV) if downsample else None
    return pcs, pd, g


@pytest.mark.parametrize("cfg", BOTTLENECK_CASES, ids=lambda c: c[0])
def test_fused_bottleneck_equals_three_launches(cfg):
    """md_bottleneck (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + residual + ReLU in one launch, intermediates in LDS) against the three
    md_conv2d launches it replaces: same operands, same K order, same bf16 rounding points -> bit-identical; and against fp32 torch."""
    from minddet_amd import _lib, nn_ops

    name, N, H, W, Cin, ds = cfg
    (pc1, pc2, pc3), pd, g = _bottleneck_modules(Cin, len(name) * 13 + H, ds)
    blk = nn_ops.pack_bottleneck(pc1, pc2, pc3, pd)
    assert blk is not None and (blk.wd is not None) == ds
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(DEV)
    res = nn_ops.conv2d(x, pd) if ds else x
    ref = nn_ops.conv2d(nn_ops.conv2d(nn_ops.conv2d(x, pc1), pc2), pc3, residual=res)
    y = nn_ops.bottleneck(x, blk)                       # identity residual, or the downsample conv computed in the launch
    assert _lib.lib().md_conv2d_last_kernel() == 7
    torch.cuda.synchronize()
    assert torch.equal(y, ref), (y.float() - ref.float()).abs().max().item()
    if ds:   # the same block with the downsample conv run by the caller (residual tensor form)
        y1 = nn_ops.bottleneck(x, nn_ops.pack_bottleneck(pc1, pc2, pc3), residual=res)
        assert torch.equal(y1, ref)
    # fp32 torch on the bf16-rounded operands (bf16 rounding of the two intermediates reproduced)
    xf = x.float().cpu().permute(0, 3, 1, 2)

    def conv(t, pc, k, pad):
        w = pc.w[:pc.cout].float().cpu().view(pc.cout, k, k, -1).permute(0, 3, 1, 2)
        return F.conv2d(t, w, pc.bias[:pc.cout].cpu(), padding=pad)

    t1 = torch.relu(conv(xf, pc1, 1, 0)).to(torch.bfloat16).float()
    t2 = torch.relu(conv(t1, pc2, 3, 1)).to(torch.bfloat16).float()
    t3 = conv(t2, pc3, 1, 0).to(torch.bfloat16).float() + res.float().cpu().permute(0, 3, 1, 2)
    t3 = torch.relu(t3).permute(0, 2, 3, 1)
    err = (y.float().cpu() - t3).abs()
    assert (err <= 2e-2 * t3.abs() + 3e-2).all(), err.max().item()


def test_fused_bottleneck_argument_checks_and_determinism():
    from minddet_amd import _lib, nn_ops

    (pc1, pc2, pc3), pd, g = _bottleneck_modules(256, 5, False)
    blk = nn_ops.pack_bottleneck(pc1, pc2, pc3)
    x = torch.randn((2, 16, 32, 256), generator=g).to(torch.bfloat16).to(DEV)
    y1, y2 = nn_ops.bottleneck(x, blk), nn_ops.bottleneck(x, blk)
    assert torch.equal(y1, y2)
    assert nn_ops.bottleneck(torch.zeros((0, 16, 32, 256), dtype=torch.bfloat16, device=DEV), blk).shape == (0, 16, 32, 256)
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):      # an identity block needs 256 input channels
        (p1, p2, p3), _, _ = _bottleneck_modules(64, 6, False)
        nn_ops.bottleneck(torch.zeros((1, 8, 16, 64), dtype=torch.bfloat16, device=DEV), nn_ops.pack_bottleneck(p1, p2, p3))
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):      # output shape
        nn_ops.bottleneck(x, blk, out=torch.zeros((2, 16, 32, 128), dtype=torch.bfloat16, device=DEV))
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):      # residual shape
        nn_ops.bottleneck(x, blk, residual=torch.zeros((2, 16, 32, 64), dtype=torch.bfloat16, device=DEV))
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):      # a fused downsample conv and a residual tensor exclude each other
        (p1, p2, p3), pd, _ = _bottleneck_modules(64, 7, True)
        b64 = nn_ops.pack_bottleneck(p1, p2, p3, pd)
        x64 = torch.zeros((1, 8, 16, 64), dtype=torch.bfloat16, device=DEV)
        _lib.call("md_bottleneck", [x64, b64.w1, b64.b12, b64.w2, b64.w3, b64.b3, torch.zeros((1, 8, 16, 256), dtype=torch.bfloat16, device=DEV),
                                    b64.wd, b64.bd, torch.zeros((1, 8, 16, 256), dtype=torch.bfloat16, device=DEV)])
    # batches whose x tensor exceeds the DMA reach run as image chunks: same result through a lowered limit (5 images -> 2 + 2 + 1)
    x5 = torch.randn((5, 16, 32, 256), generator=g).to(torch.bfloat16).to(DEV)
    y5 = nn_ops.bottleneck(x5, blk)
    y5c = nn_ops.bottleneck(x5, blk, tune=nn_ops.ConvTune(chunk_limit=2 * 16 * 32 * 256 * 2 + 1))
    assert torch.equal(y5, y5c)
    # blocks md_bottleneck does not take are not packed: the graph keeps the three-launch path for them
    p128 = nn_ops.pack_conv(torch.randn((128, 512, 1, 1)) * 0.05, relu=True).to(DEV)
    assert nn_ops.pack_bottleneck(p128, pc2, pc3) is None
    ps2 = nn_ops.pack_conv(torch.randn((64, 64, 3, 3)) * 0.05, stride=2, pad=1, relu=True).to(DEV)
    assert nn_ops.pack_bottleneck(pc1, ps2, pc3) is None


def test_resnet_with_fused_blocks_equals_layer_by_layer():
    from minddet_amd import graphs

    bb = graphs.ResNet(depth=50, base_width=64, layers=[3, 1, 1, 1], seed=9).to(DEV)
    g = torch.Generator().manual_seed(4)
    x8 = torch.zeros((2, 96, 160, 8))
    x8[..., :3] = torch.randn((2, 96, 160, 3), generator=g)
    xb = x8.to(torch.bfloat16).to(DEV)
    old, old_d = graphs.FUSE_BLOCKS, graphs.FUSE_DUAL
    try:
        graphs.FUSE_BLOCKS = graphs.FUSE_DUAL = True
        fused = bb(xb)
        assert all(b._fused is not None for b in bb.stages[0]) and all(b._fused in (None, False) for st in bb.stages[1:] for b in st)
        assert all(st[0]._dual is not None for st in bb.stages[1:])
        graphs.FUSE_BLOCKS = graphs.FUSE_DUAL = False
        plain = bb(xb)
    finally:
        graphs.FUSE_BLOCKS, graphs.FUSE_DUAL = old, old_d
    assert torch.equal(fused[0], plain[0])     # stage 1: md_bottleneck is bit-identical to the layer-by-layer launches
    for f, p in zip(fused[1:], plain[1:]):     # later stages: the first block's expand + downsample GEMM rounds once instead of three times
        assert (f.float() - p.float()).abs().max().item() <= 0.05 * max(p.float().abs().max().item(), 1.0)


@pytest.mark.parametrize("cfg", [("stage2", 2, 40, 56, 128, 256, 512, 2), ("stage3", 1, 25, 42, 256, 512, 1024, 2), ("stage4_ragged", 3, 13, 21, 512, 1024, 2048, 2),
                                 ("stride1", 2, 19, 23, 64, 128, 256, 1), ("odd_input", 1, 9, 15, 64, 64, 192, 2)], ids=lambda c: c[0])
def test_conv1x1_dual_equals_expand_plus_downsample(cfg):
    """md_conv1x1_dual ([w3 | wd] . [t2 ; x strided] + b3 + bd, ReLU) against the two md_conv2d launches it replaces (downsample conv ->
    residual of the expand conv) and against fp32 torch.  The fused form rounds the sum to bf16 once instead of three times, so the
    comparison with the layer-by-layer path allows what those roundings can move (bf16 ulps of the operands)."""
    from minddet_amd import nn_ops

    name, N, Ho, Wo, Ca, Cb, Cout, s = cfg
    g = torch.Generator().manual_seed(len(name) + Cout)
    Hb, Wb = (Ho - 1) * s + 1 + (1 if name == "stage2" else 0) * (s - 1), (Wo - 1) * s + 1 + (1 if name == "stage2" else 0) * (s - 1)
    w3 = torch.randn((Cout, Ca, 1, 1), generator=g) * (1.0 / Ca) ** 0.5
    wd = torch.randn((Cout, Cb, 1, 1), generator=g) * (1.0 / Cb) ** 0.5
    b3, bd = torch.randn((Cout,), generator=g) * 0.1, torch.randn((Cout,), generator=g) * 0.1
    pc3 = nn_ops.pack_conv(w3, bias=b3, relu=True).to(DEV)
    pd = nn_ops.pack_conv(wd, bias=bd, stride=s, relu=False).to(DEV)
    pk = nn_ops.pack_dual(pc3, pd)
    assert pk is not None and pk.stride == s
    xa = torch.randn((N, Ho, Wo, Ca), generator=g).to(torch.bfloat16).to(DEV)
    xb = torch.randn((N, Hb, Wb, Cb), generator=g).to(torch.bfloat16).to(DEV)
    ref = nn_ops.conv2d(xa, pc3, residual=nn_ops.conv2d(xb, pd))
    y = nn_ops.conv1x1_dual(xa, xb, pk)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    t = F.conv2d(xa.float().cpu().permute(0, 3, 1, 2), w3.to(torch.bfloat16).float(), None) + \
        F.conv2d(xb.float().cpu().permute(0, 3, 1, 2), wd.to(torch.bfloat16).float(), None, stride=s) + (b3 + bd).view(1, -1, 1, 1)
    t = torch.relu(t).permute(0, 2, 3, 1)
    err = (y.float().cpu() - t).abs()
    assert (err <= 2.0 ** -7 * t.abs() + 2e-2).all(), err.max().item()                 # one bf16 rounding of the exact sum (+ fp32 accumulation order)
    d = (y.float() - ref.float()).abs().cpu()
    assert (d <= 3 * 2.0 ** -7 * t.abs() + 5e-2).all(), d.max().item()                # the three roundings of the layer-by-layer path
    y2 = nn_ops.conv1x1_dual(xa, xb, pk)
    assert torch.equal(y, y2)


def test_conv1x1_dual_argument_checks():
    from minddet_amd import _lib, nn_ops

    pc3 = nn_ops.pack_conv(torch.randn((256, 64, 1, 1)) * 0.1, bias=torch.zeros(256), relu=True).to(DEV)
    pd = nn_ops.pack_conv(torch.randn((256, 128, 1, 1)) * 0.1, bias=torch.zeros(256), stride=2, relu=False).to(DEV)
    pk = nn_ops.pack_dual(pc3, pd)
    xa = torch.zeros((1, 8, 8, 64), dtype=torch.bfloat16, device=DEV)
    with pytest.raises(_lib.MindDetHipError, match="rc=2"):          # x_b's size does not give x_a's size at this stride
        nn_ops.conv1x1_dual(xa, torch.zeros((1, 20, 20, 128), dtype=torch.bfloat16, device=DEV), pk)
    with pytest.raises(_lib.MindDetHipError):                        # channel mismatch is caught by the wrapper
        nn_ops.conv1x1_dual(xa, torch.zeros((1, 15, 15, 64), dtype=torch.bfloat16, device=DEV), pk)
    assert nn_ops.conv1x1_dual(torch.zeros((0, 8, 8, 64), dtype=torch.bfloat16, device=DEV),
                               torch.zeros((0, 15, 15, 128), dtype=torch.bfloat16, device=DEV), pk).shape == (0, 8, 8, 256)
    p3x3 = nn_ops.pack_conv(torch.randn((256, 128, 3, 3)) * 0.1, stride=2, pad=1).to(DEV)
    assert nn_ops.pack_dual(pc3, p3x3) is None                       # only 1x1 downsample convs
    p64 = nn_ops.pack_conv(torch.randn((64, 64, 1, 1)) * 0.1, relu=True).to(DEV)
    assert nn_ops.pack_dual(p64, nn_ops.pack_conv(torch.randn((64, 64, 1, 1)) * 0.1).to(DEV)) is None   # Cout <= 64: other tile shape


@pytest.mark.parametrize("cfg", [
    # name, N, H, W, Cin, Cout, act, residual, x buffer channels / offset, residual buffer channels / offset, output offset
    ("k256_c1024_res", 2, 50, 84, 256, 1024, "relu", True, 0, 0, 0, 0, 0),
    ("k128_c512_res_ragged", 3, 33, 37, 128, 512, "relu", True, 0, 0, 0, 0, 0),
    ("k512_c2048_res", 1, 25, 42, 512, 2048, "relu", True, 0, 0, 0, 0, 0),
    ("k512_c128_plain", 2, 29, 31, 512, 128, "relu", False, 0, 0, 0, 0, 0),
    ("k256_c128_none", 1, 17, 19, 256, 128, "none", False, 0, 0, 0, 0, 0),
    ("k256_c256_none_res", 1, 40, 40, 256, 256, "none", True, 0, 0, 0, 0, 0),
    ("k128_c128_silu_slices", 2, 40, 40, 128, 128, "silu", True, 320, 64, 256, 128, 128),
    ("k256_c512_silu_slices", 1, 23, 45, 256, 512, "silu", True, 512, 256, 1024, 512, 64),
    ("k128_c384_relu", 1, 21, 20, 128, 384, "relu", False, 0, 0, 0, 0, 0),
    ("tiny_one_tile", 1, 3, 5, 256, 256, "relu", True, 0, 0, 0, 0, 0),
    ("one_pixel_more_than_a_tile", 1, 1, 33, 128, 256, "relu", True, 0, 0, 0, 0, 0),
], ids=lambda c: c[0])
def test_conv1x1_stream_kernel_is_bit_identical_to_the_tile_kernel(cfg):
    """conv1x1_stream_kernel (variant 30: weights in registers, activations / residual streamed) against conv_igemm_kernel
    (variant 20) on the same operands: same K order inside every MFMA chain and the same rounding points -> bit for bit;
    ragged last tiles, channel-slice inputs / residuals, concat outputs, all three activations; and against torch fp32."""
    from minddet_amd import _lib, nn_ops

    name, N, H, W, Cin, Cout, act, use_res, xc, xo, rc, ro, co = cfg
    g = torch.Generator().manual_seed(len(name) * 7 + Cin)
    wt = torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5
    pc = nn_ops.pack_conv(wt, bias=torch.randn((Cout,), generator=g) * 0.1, relu={"none": 0, "relu": 1, "silu": "silu"}[act]).to(DEV)
    xbuf = torch.randn((N, H, W, xc or Cin), generator=g).to(torch.bfloat16).to(DEV)
    rbuf = torch.randn((N, H, W, rc or Cout), generator=g).to(torch.bfloat16).to(DEV) if use_res else None
    kw = dict(x_c_off=xo if xc else None, res_c_off=ro if (rc and use_res) else None)
    last = _lib.lib().md_conv2d_last_kernel
    ref = nn_ops.conv2d(xbuf, pc, residual=rbuf, variant=20, **kw)
    assert last() != 8
    got = nn_ops.conv2d(xbuf, pc, residual=rbuf, variant=30, **kw)
    assert last() == 8, "variant 30 did not reach conv1x1_stream_kernel"
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    cat = torch.full((N, H, W, co + Cout + 8), 3.0, dtype=torch.bfloat16, device=DEV)
    kw_cat = dict(kw, res_c_off=(ro if rc else 0) if use_res else None)   # a concat output takes its residual as a channel slice
    nn_ops.conv2d(xbuf, pc, residual=rbuf, variant=30, out=cat, c_off=co, **kw_cat)
    assert last() == 8
    assert torch.equal(cat[..., co:co + Cout], ref) and (cat[..., co + Cout:] == 3.0).all() and (cat[..., :co] == 3.0).all()
    x_c = xbuf[..., xo:xo + Cin].float().cpu()
    y = F.conv2d(x_c.permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(), pc.bias[:Cout].float().cpu()).permute(0, 2, 3, 1)
    y = F.silu(y) if act == "silu" else y
    if use_res:
        r_c = rbuf[..., ro:ro + Cout].float().cpu()
        y = y.to(torch.bfloat16).float() + r_c
    y = torch.relu(y) if act == "relu" else y
    assert ((ref.float().cpu() - y).abs() <= 2e-2 * y.abs() + 2e-2).all()


@pytest.mark.parametrize("shape", [(2, 50, 84, 256, 256), (1, 25, 43, 512, 256), (3, 13, 21, 256, 512)], ids=str)
def test_conv1x1_stream_kernel_upsampled_residual(shape):
    """The FPN lateral conv with the top-down add fused (residual at half resolution, nearest 2x upsampling, odd sizes included):
    stream kernel == tile kernel bit for bit, and == conv + upsample + add in torch."""
    from minddet_amd import _lib, nn_ops

    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(H * W)
    wt = torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5
    pc = nn_ops.pack_conv(wt, bias=torch.randn((Cout,), generator=g) * 0.1, relu=False).to(DEV)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(DEV)
    r = torch.randn((N, (H + 1) // 2, (W + 1) // 2, Cout), generator=g).to(torch.bfloat16).to(DEV)
    ref = nn_ops.conv2d(x, pc, residual=r, res_upsample=True, variant=20)
    got = nn_ops.conv2d(x, pc, residual=r, res_upsample=True, variant=30)
    assert _lib.lib().md_conv2d_last_kernel() == 8
    assert torch.equal(got, ref)
    y = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(), pc.bias[:Cout].float().cpu()).permute(0, 2, 3, 1)
    up = r.float().cpu().repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)[:, :H, :W]
    y = y.to(torch.bfloat16).float() + up
    assert ((ref.float().cpu() - y).abs() <= 2e-2 * y.abs() + 2e-2).all()


def test_conv1x1_stream_kernel_many_tiles_per_workgroup_and_rounds():
    """A layer long enough that every workgroup streams many tiles (ring wrap-around, residual look-ahead), with 1 and 3
    workgroup rounds; layers the kernel does not take fall back to the dispatcher's choice under variant 30."""
    from minddet_amd import _lib, nn_ops

    g = torch.Generator().manual_seed(5)
    lib = _lib.lib()
    for (Cin, Cout) in ((256, 1024), (128, 512), (512, 256)):
        wt = torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5
        pc = nn_ops.pack_conv(wt, bias=torch.randn((Cout,), generator=g) * 0.1, relu=True).to(DEV)
        x = torch.randn((6, 100, 167, Cin), generator=g).to(torch.bfloat16).to(DEV)
        r = torch.randn((6, 100, 167, Cout), generator=g).to(torch.bfloat16).to(DEV)
        ref = nn_ops.conv2d(x, pc, residual=r, variant=20)
        for rounds in (1, 3):
            for bits in (0, 16):     # 16: the K = 512 layer on 4-wave workgroups
                got = nn_ops.conv2d(x, pc, residual=r, variant=30, tune=nn_ops.ConvTune(stream_rounds=rounds, stream_cache_bits=bits))
                assert lib.md_conv2d_last_kernel() == 8
                assert torch.equal(got, ref), (Cin, Cout, rounds, bits)
    # not pointwise / K = 64 / Cout not a multiple of 128: variant 30 is the dispatcher's own choice
    for (Cin, Cout, k) in ((64, 256, 1), (256, 64, 1), (128, 128, 3)):
        wt = torch.randn((Cout, Cin, k, k), generator=g) * 0.05
        pc = nn_ops.pack_conv(wt, pad=k // 2, relu=True).to(DEV)
        x = torch.randn((1, 20, 24, Cin), generator=g).to(torch.bfloat16).to(DEV)
        assert torch.equal(nn_ops.conv2d(x, pc, variant=30), nn_ops.conv2d(x, pc, variant=0))
        assert lib.md_conv2d_last_kernel() != 8


def test_conv1x1_stream_kernel_random_shapes_against_the_tile_kernel():
    """40 seeded random pointwise layers (ragged pixel counts from one tile to many tiles per workgroup, both K, all output widths,
    residual kinds, workgroup rounds): conv1x1_stream_kernel == conv_igemm_kernel bit for bit."""
    from minddet_amd import _lib, nn_ops

    lib = _lib.lib()
    rng = np.random.default_rng(20260101)
    g = torch.Generator().manual_seed(77)
    for case in range(40):
        cin = int(rng.choice([128, 256, 512]))
        cout = int(rng.choice([128, 256, 384, 512, 768, 1024]))
        n, h, w = int(rng.integers(1, 5)), int(rng.integers(1, 90)), int(rng.integers(1, 90))
        kind = int(rng.integers(0, 3))       # 0 none, 1 residual, 2 upsampled residual
        act = ["none", "relu", "silu"][int(rng.integers(0, 3))] if kind != 2 else "none"
        rounds = int(rng.choice([1, 2, 5]))
        wt = torch.randn((cout, cin, 1, 1), generator=g) * (2.0 / cin) ** 0.5
        pc = nn_ops.pack_conv(wt, bias=torch.randn((cout,), generator=g) * 0.1, relu={"none": 0, "relu": 1, "silu": "silu"}[act]).to(DEV)
        x = torch.randn((n, h, w, cin), generator=g).to(torch.bfloat16).to(DEV)
        r = None
        if kind == 1:
            r = torch.randn((n, h, w, cout), generator=g).to(torch.bfloat16).to(DEV)
        elif kind == 2:
            r = torch.randn((n, (h + 1) // 2, (w + 1) // 2, cout), generator=g).to(torch.bfloat16).to(DEV)
        ref = nn_ops.conv2d(x, pc, residual=r, res_upsample=kind == 2, variant=20)
        got = nn_ops.conv2d(x, pc, residual=r, res_upsample=kind == 2, variant=30, tune=nn_ops.ConvTune(stream_rounds=rounds))
        assert lib.md_conv2d_last_kernel() == 8, (case, cin, cout)
        assert torch.equal(got, ref), (case, cin, cout, n, h, w, kind, act, rounds)
        if cin == 512:   # K = 512 runs 256 couts per 8-wave workgroup where Cout % 256 == 0 (r03); the 128-cout 4-wave form stays reachable per call
            got4 = nn_ops.conv2d(x, pc, residual=r, res_upsample=kind == 2, variant=30, tune=nn_ops.ConvTune(stream_rounds=rounds, stream_cache_bits=16))
            assert lib.md_conv2d_last_kernel() == 8 and torch.equal(got4, ref), (case, cin, cout, "4-wave form")


@pytest.mark.parametrize("cfg", [
    # name, N, H, W, Cin, Cout, k, act, concat output
    ("3x3_256_many_tiles", 6, 100, 168, 256, 256, 3, "relu", False),
    ("3x3_512_two_cout_tiles_ragged", 12, 61, 57, 512, 512, 3, "relu", False),
    ("3x3_256_silu_concat", 12, 80, 80, 256, 256, 3, "silu", True),
    ("1x1_1024_256_none", 20, 50, 84, 1024, 256, 1, "none", False),
    ("16x16_mfma_form", 12, 200, 168, 256, 256, 3, "relu", False),      # M >= 400 000 pixels
    ("barely_two_rounds", 3, 150, 150, 256, 256, 3, "relu", False),     # 264 tiles on 256 workgroups
], ids=lambda c: c[0])
def test_pingpong_persistent_form_is_bit_identical(cfg):
    """The persistent form of the ping-pong kernel (variant 32: one workgroup per CU walks over several pixel tiles, next tile's
    prologue requested before the epilogue, slab epilogue) against the one-tile-per-workgroup form (variants 15 / 22): same main loop, same
    rounding -> bit for bit; partial last tiles, two cout tiles, SiLU, concat outputs, both MFMA shapes."""
    from minddet_amd import _lib, nn_ops

    name, N, H, W, Cin, Cout, k, act, cat = cfg
    g = torch.Generator().manual_seed(len(name) + Cin)
    wt = torch.randn((Cout, Cin, k, k), generator=g) * (2.0 / (k * k * Cin)) ** 0.5
    pc = nn_ops.pack_conv(wt, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=k // 2,
                          relu={"none": 0, "relu": 1, "silu": "silu"}[act]).to(DEV)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(DEV)
    # the one-tile-per-workgroup form with the MFMA shape the persistent launcher picks for this layer (15: 32x32x16, 22: 16x16x32)
    ref = nn_ops.conv2d(x, pc, variant=22 if (k == 3 and k * k * Cin >= 2304 and N * H * W >= 400000) else 15)
    assert _lib.lib().md_conv2d_last_kernel() == 1
    got = nn_ops.conv2d(x, pc, variant=32)
    assert _lib.lib().md_conv2d_last_kernel() == 1
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    if cat:
        buf = torch.full((N, H, W, Cout + 72), 3.0, dtype=torch.bfloat16, device=DEV)
        nn_ops.conv2d(x, pc, variant=32, out=buf, c_off=64)
        assert torch.equal(buf[..., 64:64 + Cout], ref) and (buf[..., :64] == 3.0).all() and (buf[..., 64 + Cout:] == 3.0).all()
    # a grid of at most one round stays on the one-tile-per-workgroup form
    xs = x[:1, :32, :32].contiguous()
    assert torch.equal(nn_ops.conv2d(xs, pc, variant=32), nn_ops.conv2d(xs, pc, variant=33))
    # spot check against fp32 on the first image
    y = F.conv2d(x[:1].float().cpu().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(), pc.bias[:Cout].float().cpu(), padding=k // 2).permute(0, 2, 3, 1)
    y = F.silu(y) if act == "silu" else (torch.relu(y) if act == "relu" else y)
    assert ((ref[:1].float().cpu() - y).abs() <= 2e-2 * y.abs() + 2e-2 * y.pow(2).mean().sqrt()).all()


@pytest.mark.parametrize("cfg", [("stage3", 24, 50, 84, 256, 512, 1024, 2), ("stage4_ragged", 48, 25, 41, 512, 1024, 2048, 2),
                                 ("stride1_long_k", 10, 61, 59, 256, 512, 512, 1)], ids=lambda c: c[0])
def test_conv1x1_dual_on_the_pingpong_kernel(cfg):
    """The long-K forms of md_conv1x1_dual (K = Ca + Cb >= 768, Cout % 256 == 0) run on the 256x256 ping-pong kernel, its K tiles past Ca
    staged from the second (strided) tensor: bit-identical to the 128x128 kernel's result (same K order, one rounding), ragged last tile."""
    from minddet_amd import _lib, nn_ops

    name, N, Ho, Wo, Ca, Cb, Cout, s = cfg
    lib = _lib.lib()
    g = torch.Generator().manual_seed(len(name) + Ca)
    Hb, Wb = (Ho - 1) * s + 1, (Wo - 1) * s + 1
    pc3 = nn_ops.pack_conv(torch.randn((Cout, Ca, 1, 1), generator=g) * (1.0 / Ca) ** 0.5, bias=torch.randn((Cout,), generator=g) * 0.1, relu=True).to(DEV)
    pd = nn_ops.pack_conv(torch.randn((Cout, Cb, 1, 1), generator=g) * (1.0 / Cb) ** 0.5, bias=torch.randn((Cout,), generator=g) * 0.1, stride=s, relu=False).to(DEV)
    pk = nn_ops.pack_dual(pc3, pd)
    xa = torch.randn((N, Ho, Wo, Ca), generator=g).to(torch.bfloat16).to(DEV)
    xb = torch.randn((N, Hb, Wb, Cb), generator=g).to(torch.bfloat16).to(DEV)
    ref = nn_ops.conv1x1_dual(xa, xb, pk, tune=nn_ops.ConvTune(dual_pp_min_k=1 << 30))
    assert lib.md_conv2d_last_kernel() == 2
    got = nn_ops.conv1x1_dual(xa, xb, pk)
    assert lib.md_conv2d_last_kernel() == 1, "the long-K dual GEMM did not reach the ping-pong kernel"
    torch.cuda.synchronize()
    assert torch.equal(got, ref)


@pytest.mark.parametrize("cfg", [
    # name, N, H, W, Cin, Cout, act, residual, concat output, channel-slice input
    ("ragged_128_256_relu", 3, 37, 53, 128, 256, "relu", False, False, False),
    ("exact_tiles_256_512_res", 2, 32, 48, 256, 512, "relu", True, False, False),
    ("odd_chunks_192_256_silu_res_cat", 2, 41, 19, 192, 256, "silu", True, True, False),
    ("one_row_of_tiles_256_256_slice_in", 1, 9, 130, 256, 256, "none", False, False, True),
    ("only_the_8x32_strip_128_256", 2, 8, 70, 128, 256, "relu", True, False, False),
    ("tiles_and_strip_256_256", 2, 40, 100, 256, 256, "relu", False, False, False),
])
def test_pingpong_halo_form_is_bit_identical(cfg):
    """The HALO form of the ping-pong kernel (16 x 16-pixel tiles, the 18 x 18 halo staged once per channel chunk, B fragments read out of it
    with shifted rows; variants 36 / 37 = 32x32x16 / 16x16x32 MFMA) == the linear-tile ping-pong kernel (variants 15 / 22) bit for bit:
    same K order, same MFMA sequence per output element.  Ragged tiles at the right / bottom image edges, an odd number of channel chunks,
    residual, SiLU, concat output, channel-slice input; and fp32 torch as the independent check."""
    from minddet_amd import _lib, nn_ops

    name, N, H, W, Cin, Cout, act, with_res, cat, slice_in = cfg
    g = torch.Generator().manual_seed(len(name) + H)
    wt = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5
    bias = torch.randn((Cout,), generator=g) * 0.1
    pc = nn_ops.pack_conv(wt, bias=bias, stride=1, pad=1, relu={"relu": 1, "silu": "silu", "none": 0}[act]).to(DEV)
    assert pc.korder == 1
    xs = 2 * Cin if slice_in else Cin
    xw = torch.randn((N, H, W, xs), generator=g).to(torch.bfloat16).to(DEV)
    x_c_off = Cin if slice_in else None
    r = torch.randn((N, H, W, Cout), generator=g).to(torch.bfloat16).to(DEV) if with_res else None
    # (a concat output takes its residual as a channel slice of a wider tensor)
    r_wide = torch.cat([torch.zeros_like(r), r], -1).contiguous() if (with_res and cat) else None

    def run(v):
        out = torch.zeros((N, H, W, 2 * Cout), dtype=torch.bfloat16, device=DEV) if cat else None
        y = nn_ops.conv2d(xw, pc, residual=r_wide if r_wide is not None else r, variant=v, out=out, c_off=Cout if cat else 0, x_c_off=x_c_off,
                          res_c_off=Cout if r_wide is not None else None)
        assert _lib.lib().md_conv2d_last_kernel() == 1
        return y

    ref0, ref1 = run(15), run(22)
    h0, h1 = run(36), run(37)
    torch.cuda.synchronize()
    assert torch.equal(ref0, ref1) and torch.equal(h0, ref0) and torch.equal(h1, ref0), name
    if cat:
        assert float(h0[..., :Cout].abs().sum()) == 0
    x = xw[..., Cin:] if slice_in else xw
    c = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wt.to(torch.bfloat16).float(), bias, padding=1)
    if act == "silu":
        c = (c * torch.sigmoid(c)).to(torch.bfloat16).float()
        if with_res:
            c = c + r.float().cpu().permute(0, 3, 1, 2)
    else:
        if with_res:
            c = c.to(torch.bfloat16).float() + r.float().cpu().permute(0, 3, 1, 2)
        if act == "relu":
            c = torch.relu(c)
    ref = c.permute(0, 2, 3, 1)
    got = (h0[..., Cout:] if cat else h0).float().cpu()
    rms = ref.pow(2).mean().sqrt().item()
    assert ((got - ref).abs() <= 1.5e-2 * ref.abs() + 1.5e-2 * rms).all(), name


def test_pingpong_halo_persistent_and_head_forms_are_bit_identical():
    """The persistent HALO form (variant 38: several 16 x 16 tiles per workgroup, slabs in the idle halo buffer, per-pixel range-checked
    stores) == the persistent linear-tile form (32); md_conv2d_head on the HALO form (variant 34) == without it (35); ragged image edges."""
    from minddet_amd import _lib, nn_ops

    g = torch.Generator().manual_seed(77)
    for (N, H, W, Cin, Cout, act) in ((6, 100, 168, 256, 256, "relu"), (5, 90, 75, 128, 512, "silu")):
        wt = torch.randn((Cout, Cin, 3, 3), generator=g) * (2.0 / (9 * Cin)) ** 0.5
        pc = nn_ops.pack_conv(wt, bias=torch.randn((Cout,), generator=g) * 0.1, stride=1, pad=1, relu=1 if act == "relu" else "silu").to(DEV)
        x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(DEV)
        ref = nn_ops.conv2d(x, pc, variant=32)
        got = nn_ops.conv2d(x, pc, variant=38)
        assert _lib.lib().md_conv2d_last_kernel() == 1
        torch.cuda.synchronize()
        assert torch.equal(got, ref) and torch.equal(ref, nn_ops.conv2d(x, pc, variant=22)), (N, H, W)
    for (n, h, w) in ((2, 96, 96), (3, 75, 91), (1, 200, 336)):
        w1 = torch.randn((256, 256, 3, 3), generator=g) * (2.0 / 2304) ** 0.5
        pc = nn_ops.pack_conv(w1, bias=torch.randn((256,), generator=g) * 0.1, stride=1, pad=1, relu=True).to(DEV)
        pc2 = nn_ops.pack_conv(torch.randn((15, 256, 1, 1), generator=g) * 0.05, bias=torch.randn((15,), generator=g) * 0.1).to(DEV)
        x = torch.randn((n, h, w, 256), generator=g).to(torch.bfloat16).to(DEV)
        a_, b_ = nn_ops.conv2d_head(x, pc, pc2, variant=35), nn_ops.conv2d_head(x, pc, pc2, variant=34)
        c_ = nn_ops.conv2d_head(x, pc, pc2)      # the dispatcher's own choice (HALO where its tiles fit): same bits either way
        torch.cuda.synchronize()
        assert torch.equal(a_, b_) and torch.equal(a_, c_), (n, h, w)


def test_repeated_bit_compare_of_the_hand_scheduled_kernels():
    """ONE test that cannot pass on a lucky run (VERDICT r03 item 7 / DESIGN 6d): md_bottleneck in its three residual modes, the persistent
    and HALO ping-pong forms, the fused head on the HALO form and the stream kernel, 3 shapes x 3 repetitions each against the kernels they
    replace, every output buffer pre-filled with a sentinel.  r03's intermittent wrong result of the block kernel showed in a loop of this shape
    every time it was run (and in a single pass only sometimes)."""
    from minddet_amd import nn_ops

    g = torch.Generator().manual_seed(77)
    bad = []

    def rep3(tag, fn, ref):
        out = torch.empty_like(ref)
        for rep in range(3):
            out.fill_(7.0)
            fn(out)
            torch.cuda.synchronize()
            if not torch.equal(out, ref):
                bad.append((tag, rep, int((out != ref).sum())))

    # md_bottleneck: fused downsample conv (MODE 2), identity residual (MODE 0), separate residual tensor (MODE 1)
    for cin, ds, ext in ((64, True, False), (256, False, False), (256, False, True)):
        (pc1, pc2, pc3), pd, g2 = _bottleneck_modules(cin, 1000 + cin + int(ext), ds)
        blk = nn_ops.pack_bottleneck(pc1, pc2, pc3, pd)
        for shape in ((1, 16, 32), (2, 24, 48), (4, 64, 64)):
            x = torch.randn(shape + (cin,), generator=g2).to(torch.bfloat16).to(DEV)
            res = nn_ops.conv2d(x, pd) if ds else (torch.randn(shape + (256,), generator=g2).to(torch.bfloat16).to(DEV) if ext else x)
            ref = nn_ops.conv2d(nn_ops.conv2d(nn_ops.conv2d(x, pc1), pc2), pc3, residual=res)
            rep3(("bottleneck", cin, ds, ext, shape), lambda o: nn_ops.bottleneck(x, blk, residual=res if ext else None, out=o), ref)
    # ping-pong forms: persistent (32), HALO 32x32x16 / 16x16x32 (36 / 37), persistent HALO (38) against the one-tile linear form (15)
    w = torch.randn((256, 256, 3, 3), generator=g) * (2.0 / 2304) ** 0.5
    pc = nn_ops.pack_conv(w, bias=torch.randn((256,), generator=g) * 0.1, stride=1, pad=1, relu=True, korder=1).to(DEV)
    pc2 = nn_ops.pack_conv(torch.randn((15, 256, 1, 1), generator=g) * 0.05, bias=torch.randn((15,), generator=g) * 0.1).to(DEV)
    for shape in ((2, 40, 56), (3, 50, 84), (1, 100, 168)):
        x = torch.randn(shape + (256,), generator=g).to(torch.bfloat16).to(DEV)
        ref = nn_ops.conv2d(x, pc, variant=15)
        for v in (32, 36, 37, 38):
            rep3(("pingpong", v, shape), lambda o, v=v: nn_ops.conv2d(x, pc, variant=v, out=o), ref)
        href = nn_ops.conv2d_head(x, pc, pc2, variant=35)
        for rep in range(3):
            got = nn_ops.conv2d_head(x, pc, pc2, variant=34)
            torch.cuda.synchronize()
            if not torch.equal(got, href):
                bad.append(("head halo", shape, rep))
    # stream kernel (30) with a residual against the tile kernel (20)
    w1 = torch.randn((1024, 256, 1, 1), generator=g) * (2.0 / 256) ** 0.5
    ps = nn_ops.pack_conv(w1, bias=torch.randn((1024,), generator=g) * 0.1, relu=True).to(DEV)
    for shape in ((2, 25, 42), (3, 50, 84), (1, 37, 53)):
        x = torch.randn(shape + (256,), generator=g).to(torch.bfloat16).to(DEV)
        r = torch.randn(shape + (1024,), generator=g).to(torch.bfloat16).to(DEV)
        ref = nn_ops.conv2d(x, ps, residual=r, variant=20)
        rep3(("stream", shape), lambda o: nn_ops.conv2d(x, ps, residual=r, variant=30, out=o), ref)
    assert not bad, bad
