"""-m gpu: md_c3_pair (the Bottleneck of a YOLOv5 C3 block in one launch: conv 1x1 + SiLU -> conv 3x3 + SiLU, + shortcut; the intermediate in
LDS) against (a) the two md_conv2d launches it replaces -- same operands, same K order, same bf16 rounding points: bit-identical -- and
(b) plain fp32 torch on the bf16-rounded operands with the intermediate rounded to bf16 as the device does.

Tolerance for (b), stated: two chained bf16 convs + one more rounding after the shortcut add -> |err| <= 2e-2 |y| + 3e-2 (the bound of the
fused ResNet block's test; measured max error 1.6e-2 on these cases)."""
import pytest
import torch
import torch.nn.functional as F

from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"

# name, N, H, W, C, shortcut, pass_through, x channels, x_c_off, y channels, y_c_off
CASES = [
    ("c64_ragged", 2, 19, 37, 64, True, False, 128, 0, 128, 0),
    ("c64_one_tile_exact", 1, 8, 16, 64, True, True, 128, 0, 128, 0),
    ("c64_smaller_than_a_tile", 1, 5, 3, 64, False, True, 128, 0, 128, 0),
    ("c64_offsets", 2, 17, 33, 64, True, False, 192, 64, 160, 96),
    ("c128_ragged", 3, 21, 50, 128, False, True, 256, 0, 256, 0),
    ("c128_one_tile_exact", 2, 8, 16, 128, True, True, 256, 0, 256, 0),
    ("c128_offsets_no_second_half", 1, 23, 19, 128, True, False, 136, 8, 128, 0),
    ("c32_ragged", 2, 19, 37, 32, True, False, 64, 0, 64, 0),
    ("c32_offsets_pass_through", 1, 9, 17, 32, False, True, 96, 32, 72, 8),
    ("c32_one_tile_exact", 3, 8, 16, 32, True, True, 64, 0, 64, 0),
    ("c32_yolov5s_p2_shard", 32, 160, 160, 32, True, True, 64, 0, 64, 0),
    ("c64_yolov5s_p3_shard", 32, 80, 80, 64, True, False, 128, 0, 128, 0),
    ("c128_yolov5s_p4_shard", 32, 40, 40, 128, True, True, 256, 0, 256, 0),
]


def _pair(C, seed):
    from minddet_amd import nn_ops

    g = torch.Generator().manual_seed(seed)
    bn = lambda c: (torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1, torch.rand(c, generator=g) + 0.5, 1e-3)
    pc1 = nn_ops.pack_conv(torch.randn((C, C, 1, 1), generator=g) * (2.0 / C) ** 0.5, bn=bn(C), relu="silu").to(DEV)
    pc2 = nn_ops.pack_conv(torch.randn((C, C, 3, 3), generator=g) * (2.0 / (9 * C)) ** 0.5, bn=bn(C), stride=1, pad=1, relu="silu").to(DEV)
    return pc1, pc2, g


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_c3_pair_equals_two_launches_and_fp32(case):
    from minddet_amd import _lib, nn_ops

    name, N, H, W, C, shortcut, passt, XC, xo, YC, yo = case
    pc1, pc2, g = _pair(C, len(name) * 7 + H)
    pk = nn_ops.pack_c3_pair(pc1, pc2)
    assert pk is not None
    x = torch.randn((N, H, W, XC), generator=g).to(torch.bfloat16).to(DEV)
    # the graph's own two-launch sequence on a copy of the slice
    t = nn_ops.conv2d(x, pc1, x_c_off=xo)
    ref = nn_ops.conv2d(t, pc2, residual=x if shortcut else None, res_c_off=xo if shortcut else None)
    y = torch.full((N, H, W, YC), 7.0, dtype=torch.bfloat16, device=DEV)
    nn_ops.c3_pair(x, pk, y, xo, yo, shortcut, passt)
    assert _lib.lib().md_conv2d_last_kernel() == 9
    torch.cuda.synchronize()
    assert torch.equal(y[..., yo:yo + C], ref), (y[..., yo:yo + C].float() - ref.float()).abs().max().item()
    # every other channel of y: the pass-through half, or untouched
    if passt:
        assert torch.equal(y[..., yo + C:yo + 2 * C], x[..., xo + C:xo + 2 * C])
        assert bool((y[..., :yo] == 7.0).all()) and bool((y[..., yo + 2 * C:] == 7.0).all())
    else:
        assert bool((y[..., :yo] == 7.0).all()) and bool((y[..., yo + C:] == 7.0).all())
    if N * H * W > 20000:
        return   # (the fp32 reference of the full shards takes a minute on the CPU: the small cases cover it)
    xf = x[..., xo:xo + C].float().cpu().permute(0, 3, 1, 2)

    def conv(tt, pc, k, pad):   # the packed weights back in torch's layout (K padded to a multiple of 64; korder 1: (ci / 64, tap, ci % 64), 0: (tap, ci))
        w = pc.w[:C, :k * k * C].float().cpu()
        if k == 1:
            w = w.view(C, C, 1, 1)
        elif getattr(pc, "korder", 0) == 1:
            w = w.view(C, C // 64, k, k, 64).permute(0, 1, 4, 2, 3).reshape(C, C, k, k)
        else:
            w = w.view(C, k, k, C).permute(0, 3, 1, 2)
        return F.conv2d(tt, w, pc.bias[:C].cpu(), padding=pad)

    t1 = F.silu(conv(xf, pc1, 1, 0)).to(torch.bfloat16).float()
    t2 = F.silu(conv(t1, pc2, 3, 1)).to(torch.bfloat16).float()
    if shortcut:
        t2 = t2 + xf
    want = t2.permute(0, 2, 3, 1)
    err = (y[..., yo:yo + C].float().cpu() - want).abs()
    assert (err <= 2e-2 * want.abs() + 3e-2).all(), err.max().item()


def test_c3_pair_argument_checks_and_determinism():
    from minddet_amd import _lib, nn_ops

    pc1, pc2, g = _pair(64, 11)
    pk = nn_ops.pack_c3_pair(pc1, pc2)
    x = torch.randn((2, 16, 32, 128), generator=g).to(torch.bfloat16).to(DEV)
    y1, y2 = torch.empty_like(x), torch.empty_like(x)
    nn_ops.c3_pair(x, pk, y1, 0, 0, True, True)
    nn_ops.c3_pair(x, pk, y2, 0, 0, True, True)
    assert torch.equal(y1, y2)
    with pytest.raises(_lib.MindDetHipError):   # in place: a tile's halo pixels are other tiles' outputs
        nn_ops.c3_pair(x, pk, x, 0, 0, True, False)
    with pytest.raises(_lib.MindDetHipError):   # the pass-through half does not fit
        nn_ops.c3_pair(x, pk, y1, 64, 0, True, True)
    with pytest.raises(_lib.MindDetHipError):   # channel offsets are multiples of 8
        nn_ops.c3_pair(x, pk, y1, 4, 0, True, False)
    with pytest.raises(_lib.MindDetHipError):   # another image size
        nn_ops.c3_pair(x, pk, torch.empty((2, 16, 31, 128), dtype=torch.bfloat16, device=DEV), 0, 0, True, False)
    # widths the kernel does not fuse are refused at pack time (the graph then keeps the two launches)
    pc1b = nn_ops.pack_conv(torch.randn((256, 256, 1, 1), generator=g), relu="silu").to(DEV)
    pc2b = nn_ops.pack_conv(torch.randn((256, 256, 3, 3), generator=g), stride=1, pad=1, relu="silu").to(DEV)
    assert nn_ops.pack_c3_pair(pc1b, pc2b) is None
    pc2r = nn_ops.pack_conv(torch.randn((64, 64, 3, 3), generator=g), stride=1, pad=1, relu=True).to(DEV)   # ReLU, not SiLU
    assert nn_ops.pack_c3_pair(pc1, pc2r) is None
    # empty batch: nothing to do, no error
    e = torch.empty((0, 16, 32, 128), dtype=torch.bfloat16, device=DEV)
    nn_ops.c3_pair(e, pk, torch.empty_like(e), 0, 0, True, False)


def test_c3_block_fused_equals_unfused():
    """graphs.C3 with its bottlenecks on md_c3_pair (the default) against the same block on two launches per bottleneck: n = 1, 2, 3
    (odd chains end in the second concat buffer and carry cv2(x) along), with and without shortcut"""
    from minddet_amd import graphs

    for n, shortcut, c2 in ((1, True, 64), (1, True, 128), (2, True, 128), (3, True, 256), (1, False, 256), (2, False, 128)):
        blk = graphs.C3(graphs.ParamInit(5 + n), c2, c2, n, shortcut)
        for m in blk.modules():
            m.to(DEV)
        x = torch.randn((2, 24, 40, c2), generator=torch.Generator().manual_seed(n)).to(torch.bfloat16).to(DEV)
        y = blk(x)
        assert blk._pairs, "the fused path did not engage"
        old = graphs.C3_PAIR_FUSED
        graphs.C3_PAIR_FUSED = False
        try:
            blk._pairs = None
            y0 = blk(x)
            assert blk._pairs is False
        finally:
            graphs.C3_PAIR_FUSED = old
            blk._pairs = None
        assert torch.equal(y, y0), (n, shortcut, c2)


@pytest.mark.parametrize("shape", [(32, 20, 20, 256), (2, 20, 20, 512), (3, 13, 17, 64), (1, 1, 1, 8), (2, 5, 40, 24), (1, 40, 40, 128)],
                         ids=lambda s_: "x".join(str(v) for v in s_))
def test_sppf_pool_equals_three_maxpools_and_torch(shape):
    """md_sppf_pool (SPPF's pooling chain, one launch, in place on the concat buffer) against three md_maxpool2d launches + copies (bit
    compare) and against torch's max_pool2d chain on the same bf16 values (max is exact: equal)."""
    from minddet_amd import _lib, nn_ops

    N, H, W, C = shape
    assert nn_ops.sppf_pool_fits(H, W, C)
    g = torch.Generator().manual_seed(H * 31 + C)
    x = torch.randn((N, H, W, C), generator=g).to(torch.bfloat16)
    for k in (5, 3):
        cat = torch.full((N, H, W, 4 * C + 8), 7.0, dtype=torch.bfloat16)
        cat[..., :C] = x
        cat = cat.to(DEV)
        nn_ops.sppf_pool(cat, C, k)
        ref = [x.to(DEV)]
        for _ in range(3):
            ref.append(nn_ops.maxpool2d(ref[-1], k, 1, k // 2, zero_pad=False))
        torch.cuda.synchronize()
        t = x.float().permute(0, 3, 1, 2)
        for i in range(4):
            assert torch.equal(cat[..., i * C:(i + 1) * C], ref[i]), (k, i)
            assert torch.equal(cat[..., i * C:(i + 1) * C].float().cpu(), t.permute(0, 2, 3, 1)), (k, i, "torch")
            t = F.max_pool2d(t, k, 1, k // 2)
        assert bool((cat[..., 4 * C:] == 7.0).all())
    # argument checks: a buffer narrower than 4 C, an even window; a map too large for LDS is refused with a size error (the graph falls back)
    with pytest.raises(_lib.MindDetHipError):
        nn_ops.sppf_pool(torch.empty((1, 4, 4, 3 * C), dtype=torch.bfloat16, device=DEV), C, 5)
    with pytest.raises(_lib.MindDetHipError):
        nn_ops.sppf_pool(torch.empty((1, 4, 4, 4 * C), dtype=torch.bfloat16, device=DEV), C, 4)
    assert not nn_ops.sppf_pool_fits(80, 80, 64)
    with pytest.raises(_lib.MindDetHipError):
        nn_ops.sppf_pool(torch.empty((1, 80, 80, 256), dtype=torch.bfloat16, device=DEV), 64, 5)


def test_sppf_block_fused_equals_unfused():
    from minddet_amd import graphs

    blk = graphs.SPPF(graphs.ParamInit(3), 128, 128)
    for m in blk.modules():
        m.to(DEV)
    x = torch.randn((2, 20, 24, 128), generator=torch.Generator().manual_seed(1)).to(torch.bfloat16).to(DEV)
    y = blk(x)
    old = graphs.SPPF_FUSED
    graphs.SPPF_FUSED = False
    try:
        y0 = blk(x)
    finally:
        graphs.SPPF_FUSED = old
    assert torch.equal(y, y0)
