"""-m gpu: the reference-present graphs -- CenterNet-R18 (neck with Conv2dTranspose 4x4 s2 p1, fused heads,
sigmoid+clip, max-pool NMS, two-stage top-k, gather decode) and the CenterPoint RPN neck (strided conv /
Conv2dTranspose k=s deblocks writing a channel-concatenated output) -- against the torch/numpy oracle.
Conv stacks: bf16 fp tolerance; decode indices: bit-exact from the device head tensors."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nets, np_ops
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


@pytest.mark.parametrize("k,s,p,cin,cout", [(4, 2, 1, 64, 64), (2, 2, 0, 128, 128), (4, 4, 0, 256, 128), (4, 2, 1, 16, 24)])
def test_conv_transpose_vs_torch(k, s, p, cin, cout):
    from minddet_amd import nn_ops

    g = torch.Generator().manual_seed(k * 10 + s)
    wt = torch.randn((cin, cout, k, k), generator=g) * (2.0 / (k * k * cin)) ** 0.5
    bn = (torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1, torch.randn(cout, generator=g) * 0.1,
          torch.rand(cout, generator=g) + 0.5, 1e-3)
    pct = nn_ops.pack_conv_transpose(wt, bn=bn, stride=s, pad=p, relu=True).to(DEV)
    x = torch.randn((2, 13, 17, cin), generator=g).to(torch.bfloat16)
    y = nn_ops.conv_transpose2d(x.to(DEV), pct).float().cpu()
    scale = bn[0] / torch.sqrt(bn[3] + bn[4])
    wf = (wt * scale.view(1, -1, 1, 1)).to(torch.bfloat16).float()
    ref = torch.relu(F.conv_transpose2d(x.float().permute(0, 3, 1, 2), wf, bn[1] - bn[2] * scale, stride=s, padding=p))
    ref = ref.permute(0, 2, 3, 1)
    assert y.shape == ref.shape == (2, 13 * s, 17 * s, cout)
    rms = ref.pow(2).mean().sqrt().item()
    assert ((y - ref).abs() <= 1.2e-2 * ref.abs() + 1.2e-2 * rms).all()


def test_centerpoint_rpn_neck_shape_and_values():
    from minddet_amd import graphs

    # known answer from the reference's own smoke test (rpn.py:157-164): [4,64,512,512] -> [4,384,128,128];
    # run here at 1/4 size so that the fp32 oracle finishes in seconds: [1,64,128,128] -> [1,384,32,32]
    neck = graphs.RPN(num_input_features=64).to(DEV)
    assert neck.out_channels == 384
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 128, 128, 64), generator=g).to(torch.bfloat16)
    y = neck(x.to(DEV)).float().cpu()
    assert y.shape == (1, 32, 32, 384)
    ref = nets.rpn_neck_forward(neck, x.float().permute(0, 3, 1, 2), quant=True).permute(0, 2, 3, 1)
    rms = ref.pow(2).mean().sqrt().item()
    err = (y - ref).abs().max().item()
    assert err <= 5e-2 * (rms + ref.abs().max().item() * 0.1), (err, rms)
    # full reference size: shape only
    y4 = neck(torch.zeros((4, 512, 512, 64), dtype=torch.bfloat16, device=DEV))
    assert y4.shape == (4, 128, 128, 384)


def test_centernet_end_to_end():
    from minddet_amd import graphs

    m = graphs.CenterNet(depth=18, num_classes=80).to(DEV)
    g = torch.Generator().manual_seed(1)
    x = torch.zeros((1, 256, 320, 8))
    x[..., :3] = torch.randn((1, 256, 320, 3), generator=g)
    xb = x.to(torch.bfloat16)
    det, aux = m.forward(xb.to(DEV), return_aux=True)
    torch.cuda.synchronize()
    assert det.shape == (1, 100, 6) and aux["hm"].shape == (1, 80, 64, 80)
    # heads vs the UNFUSED fp32 oracle (checks the fused 3x3 / block-diagonal 1x1 forms too)
    ref = nets.centernet_features(m, xb[..., :3].float().permute(0, 3, 1, 2).contiguous(), quant=True)
    head = aux["head"].float().cpu().permute(0, 3, 1, 2)
    for name, sl in (("hm", slice(0, 80)), ("wh", slice(80, 82)), ("reg", slice(82, 84))):
        r = ref[name]
        err = (head[:, sl] - r).abs().max().item()
        assert err <= 5e-2 * (1 + r.abs().max().item()), (name, err)
    # decode from the DEVICE head tensors: indices / classes bit-exact, boxes to fp tolerance
    hm, wh, reg = aux["hm"].cpu().numpy(), aux["wh"].cpu().numpy(), aux["reg"].cpu().numpy()
    d_o, i_o, c_o = np_ops.centernet_decode(hm, wh, reg, 100)
    np.testing.assert_array_equal(aux["inds"].cpu().numpy(), i_o)
    np.testing.assert_array_equal(aux["cls"].cpu().numpy(), c_o)
    np.testing.assert_allclose(det.cpu().numpy(), d_o, rtol=1e-6, atol=1e-5)
    # sigmoid+clip bounds (utils.py:132-157)
    assert hm.min() >= 1e-4 - 1e-9 and hm.max() <= 1 - 1e-4 + 1e-7
