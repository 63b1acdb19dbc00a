"""-m gpu: modulated deformable conv (DCNv2: md_deform_cols + md_conv2d) against the torch-CPU oracle of the published
definition (oracle/nets.py::deform_conv_module; the MindSpore primitive's arithmetic is not in the reference: parity
unpinned).  Tolerance: bf16 data path on both sides (offsets, columns and output rounded to bf16): rtol/atol 2e-2 of rms --
a bf16 offset that lands within 2^-8 of an integer coordinate may pick the neighbouring pixel pair with ~zero weight on one
side and not the other, which moves the sample by at most one bf16 ulp of the column."""
import pytest
import torch

from oracle import nets
from tests.conftest import has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs MI355X")]
DEV = "cuda:0"


@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 128), (1, 13, 21, 128, 64), (1, 32, 32, 512, 256)])
def test_deform_conv_vs_oracle(shape):
    from minddet_amd import graphs

    n, h, w, cin, cout = shape
    m = graphs.DeformConvModule(graphs.ParamInit(3 + h), cin, cout, 3, 1, 1, offset_std=0.05).to(DEV)
    g = torch.Generator().manual_seed(h)
    x = torch.randn((n, h, w, cin), generator=g).to(torch.bfloat16)
    y = m(x.to(DEV)).float().cpu().permute(0, 3, 1, 2)
    ref = nets.deform_conv_module(m, x.float().permute(0, 3, 1, 2).contiguous(), quant=True)
    assert y.shape == ref.shape
    rms = ref.pow(2).mean().sqrt().item()
    err = (y - ref).abs()
    assert (err <= 2e-2 * ref.abs() + 2e-2 * rms).float().mean() > 0.999, err.max().item()
    assert err.max().item() <= 0.25 * rms + 0.1 * ref.abs().max().item()


def test_zero_offsets_reduce_to_half_a_plain_conv():
    """Offset conv weights and biases zero (the reference's initialisation): every tap samples its integer position with
    mask sigmoid(0) = 0.5, so the layer equals 0.5 x the plain 3x3 conv."""
    from minddet_amd import graphs, nn_ops

    m = graphs.DeformConvModule(graphs.ParamInit(1), 64, 64, 3, 1, 1, bn=False, relu=False)
    m.offset_weight.zero_()
    m.offset_bias.zero_()
    m.to(DEV)
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, 12, 20, 64), generator=g).to(torch.bfloat16).to(DEV)
    y = m(x).float().cpu()
    pc = nn_ops.pack_conv(m.weight * 0.5, stride=1, pad=1).to(DEV)
    ref = nn_ops.conv2d(x, pc).float().cpu()
    rms = ref.pow(2).mean().sqrt().item()
    assert ((y - ref).abs() <= 1.2e-2 * ref.abs() + 1.2e-2 * rms).all()
