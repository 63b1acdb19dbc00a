"""Host side of the conv / pooling / resampling kernels (NHWC bf16 activations).

The packing functions turn the reference's parameter layout -- Conv2d weight [Cout,Cin,kh,kw]
+ BatchNorm2d (gamma, beta, moving_mean, moving_variance, eps), as in
minddet/models/centernet/src/resnet.py:109-178 -- into the folded, K-major bf16 blobs
md_conv2d consumes (w' = w*gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps)).
Packing is one-time model-build work and runs in torch; the per-step path is HIP only.
"""
import ctypes

import torch

from . import _lib


class _ConvAttrs(ctypes.Structure):
    _fields_ = [("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("relu", ctypes.c_int32), ("variant", ctypes.c_int32)]


def cout_tile(cout):
    return 128 if cout > 64 else (64 if cout > 32 else 32)


def _round_up(x, m):
    return (x + m - 1) // m * m


class PackedConv:
    """Folded conv weights in the kernel's layout + the static attributes of the layer."""

    def __init__(self, w, bias, cin, cout, kh, kw, stride, pad, relu):
        self.w, self.bias = w, bias
        self.cin, self.cout, self.kh, self.kw = cin, cout, kh, kw
        self.stride, self.pad, self.relu = stride, pad, relu

    def to(self, device):
        self.w, self.bias = self.w.to(device), self.bias.to(device)
        return self

    def flops(self, n, ho, wo):
        return 2 * n * ho * wo * self.cout * self.cin_real * self.kh * self.kw

    cin_real = 0


def pack_conv(weight, bias=None, bn=None, stride=1, pad=0, relu=False, cin_pad_to=8, cout_pad_to=8):
    """weight [Cout,Cin,kh,kw] fp32 (torch, any device). bn = (gamma, beta, mean, var, eps) or None."""
    weight = weight.detach().to(torch.float32)
    cout, cin, kh, kw = weight.shape
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        scale = gamma.to(torch.float32) / torch.sqrt(var.to(torch.float32) + eps)
        weight = weight * scale.view(-1, 1, 1, 1)
        b = beta.to(torch.float32) - mean.to(torch.float32) * scale
        if bias is not None:
            b = b + bias.to(torch.float32) * scale
    else:
        b = bias.detach().to(torch.float32) if bias is not None else torch.zeros(cout)
    cin_p = _round_up(cin, cin_pad_to)
    cout_o = _round_up(cout, cout_pad_to)           # channels the output tensor carries
    cout_p = _round_up(cout_o, cout_tile(cout_o))    # rows of the packed weight
    k_real = kh * kw * cin_p
    k_pad = _round_up(k_real, 64)
    wp = torch.zeros((cout_p, kh, kw, cin_p), dtype=torch.float32, device=weight.device)
    wp[:cout, :, :, :cin] = weight.permute(0, 2, 3, 1)
    wk = torch.zeros((cout_p, k_pad), dtype=torch.float32, device=weight.device)
    wk[:, :k_real] = wp.reshape(cout_p, k_real)
    bp = torch.zeros((cout_p,), dtype=torch.float32, device=weight.device)
    bp[:cout] = b.to(weight.device)
    pc = PackedConv(wk.to(torch.bfloat16).contiguous(), bp.contiguous(), cin_p, cout_o, kh, kw, stride, pad, relu)
    pc.cin_real = cin
    return pc


def conv_out_hw(h, w, pc):
    return (h + 2 * pc.pad - pc.kh) // pc.stride + 1, (w + 2 * pc.pad - pc.kw) // pc.stride + 1


CONV_VARIANT = 0  # 0 auto; 1/2/3 force a kernel variant (A/B measurements, see md_conv2d_attrs)


def conv2d(x, pc, residual=None, relu=None, out=None, variant=None):
    """x [N,H,W,Cin] bf16 NHWC contiguous CUDA tensor -> y [N,Ho,Wo,Cout] bf16."""
    n, h, w, c = x.shape
    if c != pc.cin:
        raise _lib.MindDetHipError(f"conv2d: input has {c} channels, layer packed for {pc.cin}")
    ho, wo = conv_out_hw(h, w, pc)
    if out is None:
        out = torch.empty((n, ho, wo, pc.cout), dtype=torch.bfloat16, device=x.device)
    attrs = _ConvAttrs(pc.kh, pc.kw, pc.stride, pc.pad, int(pc.relu if relu is None else relu),
                       int(CONV_VARIANT if variant is None else variant))
    _lib.call("md_conv2d", [x, pc.w, pc.bias, residual, out], extra=attrs)
    return out


def linear(x2d, pc, relu=None):
    """FC as a 1x1 conv: x2d [R, Cin] bf16 -> [R, Cout] bf16."""
    r, c = x2d.shape
    y = conv2d(x2d.view(r, 1, 1, c), pc, relu=relu)
    return y.view(r, pc.cout)


class _PoolAttrs(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("zero_pad", ctypes.c_int32)]


class _SliceAttrs(ctypes.Structure):
    _fields_ = [("c0", ctypes.c_int32), ("width", ctypes.c_int32)]


def maxpool2d(x, k=3, stride=2, pad=1, zero_pad=True):
    """NHWC bf16 max-pool. zero_pad=True is the reference stem (explicit zero Pad then
    MaxPool2d(3,2), centernet/src/resnet.py:199-204)."""
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = torch.empty((n, ho, wo, c), dtype=torch.bfloat16, device=x.device)
    _lib.call("md_maxpool2d", [x, y], extra=_PoolAttrs(k, stride, pad, int(zero_pad)))
    return y


def upsample_add(lateral, top):
    """FPN top-down: lateral + nearest-upsampled top (to lateral's size)."""
    y = torch.empty_like(lateral)
    _lib.call("md_upsample_add", [lateral, top, y])
    return y


def slice_cast(x, c0, width):
    """x[..., c0:c0+width] (bf16) -> fp32 contiguous."""
    y = torch.empty(tuple(x.shape[:-1]) + (width,), dtype=torch.float32, device=x.device)
    _lib.call("md_slice_cast", [x, y], extra=_SliceAttrs(c0, width))
    return y
