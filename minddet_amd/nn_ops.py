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


class ConvTune(ctypes.Structure):
    """md_conv_tune: per-call tuning knobs of the conv family (all zero = the library's defaults; nothing persists in the library)."""
    _fields_ = [("chunk_limit", ctypes.c_int32), ("stream_rounds", ctypes.c_int32), ("stream_wgs_per_cu", ctypes.c_int32),
                ("stream_cache_bits", ctypes.c_int32), ("pers_min_k", ctypes.c_int32), ("dual_pp_min_k", ctypes.c_int32)]


class _ConvAttrs(ctypes.Structure):
    _fields_ = [("kh", ctypes.c_int32), ("kw", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32),
                ("relu", ctypes.c_int32), ("variant", ctypes.c_int32), ("adv", ctypes.c_int32), ("pad_top", ctypes.c_int32),
                ("pad_left", ctypes.c_int32), ("sub_h", ctypes.c_int32), ("sub_w", ctypes.c_int32),
                ("out_stride", ctypes.c_int32), ("out_off_y", ctypes.c_int32), ("out_off_x", ctypes.c_int32),
                ("c_off", ctypes.c_int32), ("cout", ctypes.c_int32), ("res_upsample", ctypes.c_int32), ("korder", ctypes.c_int32),
                ("x_c_off", ctypes.c_int32), ("x_cin", ctypes.c_int32), ("res_slice", ctypes.c_int32), ("res_c_off", ctypes.c_int32),
                ("reserved0", ctypes.c_int32), ("tune", ConvTune)]


def cout_tile(cout):
    return 128 if cout > 64 else (64 if cout > 32 else 32)


def _round_up(x, m):
    return (x + m - 1) // m * m


KORDER_DEFAULT = 0  # K order for multi-tap convs with Cin % 64 == 0 (see md_conv2d_attrs.korder)


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


def pack_conv(weight, bias=None, bn=None, stride=1, pad=0, relu=False, cin_pad_to=8, cout_pad_to=8, korder=None):
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
    if korder is None:
        # 3x3 / stride 1 / pad 1 on >= 64 channels with >= 72 output channels runs on the halo-reuse kernel, which
        # wants the (ci/64, tap, ci%64) K order; everything else keeps (tap, ci)
        halo = kh == 3 and kw == 3 and stride == 1 and pad == 1 and cin_p % 64 == 0 and cout_o >= 64
        korder = 1 if halo else KORDER_DEFAULT
    if korder == 1:  # (ci/64, kh, kw, ci%64)
        wp = wp.reshape(cout_p, kh, kw, cin_p // 64, 64).permute(0, 3, 1, 2, 4).contiguous()
    wk = torch.zeros((cout_p, k_pad), dtype=torch.float32, device=weight.device)
    wk[:, :k_real] = wp.reshape(cout_p, k_real)
    bp = torch.zeros((cout_p,), dtype=torch.float32, device=weight.device)
    bp[:cout] = b.to(weight.device)
    act = {None: 0, False: 0, True: 1, "relu": 1, "silu": 2, 0: 0, 1: 1, 2: 2}[relu]  # md_conv2d_attrs.relu codes
    pc = PackedConv(wk.to(torch.bfloat16).contiguous(), bp.contiguous(), cin_p, cout_o, kh, kw, stride, pad, act)
    pc.cin_real = cin
    pc.korder = korder
    return pc


def conv_out_hw(h, w, pc):
    return (h + 2 * pc.pad - pc.kh) // pc.stride + 1, (w + 2 * pc.pad - pc.kw) // pc.stride + 1


import os as _os
# A/B knobs (tools): MD_DUAL_PP_MIN_K = concatenated K from which md_conv1x1_dual runs on the ping-pong kernel, MD_PERS_MIN_K = K from
# which the persistent ping-pong form is the dispatcher's choice.  They travel in every call's attribute struct (md_conv_tune).
TUNE = ConvTune(dual_pp_min_k=int(_os.environ.get("MD_DUAL_PP_MIN_K", "0")), pers_min_k=int(_os.environ.get("MD_PERS_MIN_K", "0")))
CONV_VARIANT = int(_os.environ.get("MD_CONV_VARIANT", "0"))  # 0 auto; other values force a kernel variant (A/B measurements, see md_conv2d_attrs; 31 = auto without conv1x1_stream_kernel)


def conv2d(x, pc, residual=None, relu=None, out=None, variant=None, c_off=0, res_upsample=False, x_c_off=None, res_c_off=None, tune=None):
    """x [N,H,W,Cin] bf16 NHWC contiguous CUDA tensor -> y [N,Ho,Wo,Cout] bf16.
    With `out` wider than the layer (channel concat, rpn.py:152) the result goes to channels
    [c_off, c_off + Cout) of `out`.  x_c_off: the layer reads channels [x_c_off, x_c_off + pc.cin) of a wider x;
    res_c_off: the residual is channels [res_c_off, res_c_off + Cout) of a wider [N,Ho,Wo,R] tensor."""
    n, h, w, c = x.shape
    if x_c_off is None and c != pc.cin:
        raise _lib.MindDetHipError(f"conv2d: input has {c} channels, layer packed for {pc.cin}")
    ho, wo = conv_out_hw(h, w, pc)
    if out is None:
        out = torch.empty((n, ho, wo, pc.cout), dtype=torch.bfloat16, device=x.device)
    attrs = _ConvAttrs(pc.kh, pc.kw, pc.stride, pc.pad, int(pc.relu if relu is None else relu),
                       int(CONV_VARIANT if variant is None else variant))
    attrs.korder = getattr(pc, "korder", 0)
    attrs.tune = TUNE if tune is None else tune
    attrs.res_upsample = int(bool(res_upsample))
    if x_c_off is not None:
        attrs.x_c_off, attrs.x_cin = int(x_c_off), pc.cin
    if res_c_off is not None:
        attrs.res_slice, attrs.res_c_off = 1, int(res_c_off)
    if out.shape[3] != pc.cout or c_off:
        attrs.adv, attrs.pad_top, attrs.pad_left, attrs.sub_h, attrs.sub_w = 1, pc.pad, pc.pad, ho, wo
        attrs.out_stride, attrs.c_off, attrs.cout = 1, int(c_off), pc.cout
    _lib.call("md_conv2d", [x, pc.w, pc.bias, residual, out], extra=attrs)
    return out


def conv2d_head(x, pc, pc2, variant=None, tune=None):
    """conv (256 output channels, ReLU) + 1x1 head with <= 16 output channels in one md_conv2d_head call: [N,H,W,Cin] ->
    [N,Ho,Wo,16] bf16 (the RPN head).  Falls back to two md_conv2d launches inside the library where the fused kernel
    does not apply."""
    n, h, w, c = x.shape
    if c != pc.cin or pc.cout != 256 or pc2.cin != 256 or pc2.cout != 16 or pc2.kh != 1 or pc2.stride != 1 or pc2.pad != 0:
        raise _lib.MindDetHipError("conv2d_head: needs a 256-channel conv followed by a 1x1 conv with 16 (padded) output channels")
    ho, wo = conv_out_hw(h, w, pc)
    y2 = torch.empty((n, ho, wo, 16), dtype=torch.bfloat16, device=x.device)
    attrs = _ConvAttrs(pc.kh, pc.kw, pc.stride, pc.pad, 1, int(CONV_VARIANT if variant is None else variant))
    attrs.korder = getattr(pc, "korder", 0)
    attrs.tune = TUNE if tune is None else tune
    _lib.call("md_conv2d_head", [x, pc.w, pc.bias, pc2.w, pc2.bias, y2], extra=attrs)
    return y2


class PackedBottleneck:
    """The three packed convs of a stride-1 bottleneck block with 64 mid channels (+ its 1x1 downsample conv), as md_bottleneck
    consumes them."""

    def __init__(self, pc1, pc2, pc3, pd=None):
        self.cin, self.cout = pc1.cin, pc3.cout
        self.w1, self.w2, self.w3 = pc1.w, pc2.w, pc3.w
        self.b12 = torch.cat([pc1.bias[:64], pc2.bias[:64]]).contiguous()
        self.b3 = pc3.bias
        self.wd, self.bd = (pd.w, pd.bias) if pd is not None else (None, None)
        self.macs_per_pixel = pc1.cin_real * 64 + 9 * 64 * 64 + 64 * 256 + (pd.cin_real * 256 if pd is not None else 0)

    def flops_bytes(self, n, h, w, with_residual_tensor=False):
        """algorithmic flops and bytes of the block as ONE op: x read once, y written once (+ a separate residual tensor)"""
        px = n * h * w
        wts = self.w1.numel() + self.w2.numel() + self.w3.numel() + (self.wd.numel() if self.wd is not None else 0)
        byts = 2.0 * (px * (self.cin + 256 + (256 if with_residual_tensor else 0)) + wts)
        return 2.0 * px * self.macs_per_pixel, byts


def pack_bottleneck(pc1, pc2, pc3, pd=None):
    """-> PackedBottleneck if (conv1 1x1 -> conv2 3x3 -> conv3 1x1 + residual) is the shape md_bottleneck fuses, else None.
    pd: the block's downsample conv; it is fused too when it is a plain 1x1 / stride 1 conv on 64 input channels (then
    bottleneck() needs no residual argument), otherwise the caller runs it and passes its output as the residual."""
    ok = (pc1.kh == 1 and pc1.stride == 1 and pc1.pad == 0 and pc1.relu == 1 and pc1.cout == 64 and pc1.cin in (64, 256) and
          tuple(pc1.w.shape) == (64, pc1.cin) and
          pc2.kh == 3 and pc2.kw == 3 and pc2.stride == 1 and pc2.pad == 1 and pc2.relu == 1 and pc2.cin == 64 and pc2.cout == 64 and
          tuple(pc2.w.shape) == (64, 576) and
          pc3.kh == 1 and pc3.stride == 1 and pc3.pad == 0 and pc3.relu == 1 and pc3.cin == 64 and pc3.cout == 256 and
          tuple(pc3.w.shape) == (256, 64))
    if not ok:
        return None
    fuse_ds = (pd is not None and pd.kh == 1 and pd.stride == 1 and pd.pad == 0 and pd.relu == 0 and pd.cin == 64 and pc1.cin == 64 and
               pd.cout == 256 and tuple(pd.w.shape) == (256, 64))
    return PackedBottleneck(pc1, pc2, pc3, pd if fuse_ds else None)


def bottleneck(x, blk, residual=None, out=None, tune=None):
    """y = relu(conv3(relu(conv2(relu(conv1(x))))) + residual) in one md_bottleneck launch.  residual None: the block's own
    downsample conv if it was packed into blk (computed in the launch), else x itself (identity block)."""
    n, h, w, c = x.shape
    if c != blk.cin:
        raise _lib.MindDetHipError(f"bottleneck: input has {c} channels, block packed for {blk.cin}")
    if out is None:
        out = torch.empty((n, h, w, 256), dtype=torch.bfloat16, device=x.device)
    fused_ds = blk.wd is not None and residual is None
    _lib.call("md_bottleneck", [x, blk.w1, blk.b12, blk.w2, blk.w3, blk.b3, residual, blk.wd if fused_ds else None,
                                blk.bd if fused_ds else None, out], extra=tune)
    return out


class _C3PairAttrs(ctypes.Structure):
    _fields_ = [("x_c_off", ctypes.c_int32), ("y_c_off", ctypes.c_int32), ("shortcut", ctypes.c_int32), ("pass_through", ctypes.c_int32)]


class PackedC3Pair:
    """The (1x1, 3x3) conv pair of a YOLOv5 C3 Bottleneck as md_c3_pair consumes it."""

    def __init__(self, pc1, pc2):
        self.c = pc1.cin
        self.w1, self.w2 = pc1.w, pc2.w
        self.b12 = torch.cat([pc1.bias[:self.c], pc2.bias[:self.c]]).contiguous()
        self.macs_per_pixel = pc1.cin_real * self.c + 9 * pc2.cin_real * self.c

    def flops_bytes(self, n, h, w, shortcut=True):
        """algorithmic flops and bytes of the pair as ONE op: x read once, y written once"""
        px = n * h * w
        return 2.0 * px * self.macs_per_pixel, 2.0 * (px * 2 * self.c + self.w1.numel() + self.w2.numel())


def pack_c3_pair(pc1, pc2):
    """-> PackedC3Pair if (conv 1x1 + SiLU -> conv 3x3 / s1 / p1 + SiLU) on C = 32, 64 or 128 channels is the shape md_c3_pair fuses, else None"""
    c = pc1.cin
    k1, k2 = _round_up(c, 64), _round_up(9 * c, 64)
    ok = (c in (32, 64, 128) and pc1.kh == 1 and pc1.stride == 1 and pc1.pad == 0 and pc1.relu == 2 and pc1.cout == c and tuple(pc1.w.shape) == (c, k1) and
          pc2.kh == 3 and pc2.kw == 3 and pc2.stride == 1 and pc2.pad == 1 and pc2.relu == 2 and pc2.cin == c and pc2.cout == c and
          tuple(pc2.w.shape) == (c, k2) and getattr(pc2, "korder", 0) == (0 if c == 32 else 1))
    return PackedC3Pair(pc1, pc2) if ok else None


def c3_pair(x, pk, out, x_c_off=0, y_c_off=0, shortcut=True, pass_through=False):
    """out[..., y_c_off : +C] = (x[..., x_c_off : +C] if shortcut) + silu(conv3x3(silu(conv1x1(x[..., x_c_off : +C])))) in one md_c3_pair launch;
    pass_through: out[..., y_c_off + C : + 2C] = x[..., x_c_off + C : + 2C].  `out` must be another buffer than x."""
    _lib.call("md_c3_pair", [x, pk.w1, pk.b12, pk.w2, out], extra=_C3PairAttrs(int(x_c_off), int(y_c_off), int(bool(shortcut)), int(bool(pass_through))))
    return out


class _DualAttrs(ctypes.Structure):
    _fields_ = [("stride_b", ctypes.c_int32), ("relu", ctypes.c_int32), ("tune", ConvTune)]


class PackedDual:
    """[w3 | wd] and b3 + bd of a bottleneck's expand conv and its 1x1 downsample conv (md_conv1x1_dual)."""

    def __init__(self, pc3, pd):
        self.ca, self.cb, self.cout, self.stride, self.relu = pc3.cin, pd.cin, pc3.cout, pd.stride, int(pc3.relu)
        self.w = torch.cat([pc3.w[:, :pc3.cin], pd.w[:, :pd.cin]], dim=1).contiguous()
        self.bias = (pc3.bias + pd.bias).contiguous()
        self.cin_real = pc3.cin_real + pd.cin_real

    def flops_bytes(self, n, ho, wo):
        px = n * ho * wo
        return 2.0 * px * self.cout * (self.ca + self.cb), 2.0 * (px * (self.ca + self.cb + self.cout) + self.w.numel())


def pack_dual(pc3, pd):
    """-> PackedDual when conv3 (1x1, stride 1) and the downsample conv (1x1, stride s, no activation) can run as one md_conv1x1_dual
    GEMM, else None."""
    ok = (pc3.kh == 1 and pc3.stride == 1 and pc3.pad == 0 and pc3.relu in (0, 1) and pd.kh == 1 and pd.pad == 0 and pd.relu == 0 and
          pc3.cout == pd.cout and pc3.cout > 64 and pc3.cin % 64 == 0 and pd.cin % 64 == 0 and cout_tile(pc3.cout) == 128 and
          tuple(pc3.w.shape) == (pd.w.shape[0], pc3.cin) and pd.w.shape[1] == pd.cin)
    return PackedDual(pc3, pd) if ok else None


def conv1x1_dual(xa, xb, pk, out=None, tune=None):
    """y = act(w3 . xa + wd . xb[:, ::s, ::s] + b3 + bd) in one launch (xa [N,Ho,Wo,Ca], xb [N,Hb,Wb,Cb])."""
    n, ho, wo, ca = xa.shape
    if ca != pk.ca or xb.shape[3] != pk.cb:
        raise _lib.MindDetHipError(f"conv1x1_dual: inputs have {ca} / {xb.shape[3]} channels, packed for {pk.ca} / {pk.cb}")
    if out is None:
        out = torch.empty((n, ho, wo, pk.cout), dtype=torch.bfloat16, device=xa.device)
    _lib.call("md_conv1x1_dual", [xa, xb, pk.w, pk.bias, None, out], extra=_DualAttrs(int(pk.stride), int(pk.relu), TUNE if tune is None else tune))
    return out


class PackedConvT:
    """A transposed conv as s*s sub-pixel convs on the MFMA kernel (one launch per output parity)."""

    def __init__(self, subs, cin, cout, k, stride, relu):
        self.subs, self.cin, self.cout, self.k, self.stride, self.relu = subs, cin, cout, k, stride, relu

    def to(self, device):
        for pc, _ in self.subs:
            pc.to(device)
        return self


def pack_conv_transpose(weight_t, bias=None, bn=None, stride=2, pad=1, relu=False):
    """weight_t [Cin,Cout,k,k] (MindSpore/PyTorch Conv2dTranspose layout).  Supported: (k=4,s=2,p=1) --
    centernet/src/centernet_det.py:145-152 -- and (k=s, p=0) -- centerpoint/det3d_ms/models/necks/rpn.py:66-80.
    Output pixel oy = iy*s - p + ky  =>  for output parity py the contributing kernel rows are
    ky = (py + p) mod s, +s, ... each paired with input row iy = (oy + p - ky) / s."""
    cin, cout, k, k2 = weight_t.shape
    s = stride
    if not (k == k2 and ((k == 4 and s == 2 and pad == 1) or (k == s and pad == 0))):
        raise _lib.MindDetHipError("pack_conv_transpose: unsupported (kernel, stride, padding)")
    subs = []
    for py in range(s):
        for px in range(s):
            kys = [ky for ky in range(k) if (ky - py - pad) % s == 0]
            kxs = [kx for kx in range(k) if (kx - px - pad) % s == 0]
            # input row for output row oy = s*i + py and kernel row ky: iy = i + (py + pad - ky)/s ; order taps by iy
            dys = sorted(((py + pad - ky) // s, ky) for ky in kys)
            dxs = sorted(((px + pad - kx) // s, kx) for kx in kxs)
            w_sub = torch.stack([torch.stack([weight_t[:, :, ky, kx] for _, kx in dxs], -1) for _, ky in dys], -2)
            # [Cin, Cout, nty, ntx] -> conv layout [Cout, Cin, nty, ntx]
            pc = pack_conv(w_sub.permute(1, 0, 2, 3).contiguous(), bias=bias, bn=bn, stride=1, pad=0, relu=relu)
            subs.append((pc, dict(py=py, px=px, pad_top=-dys[0][0], pad_left=-dxs[0][0])))
    return PackedConvT(subs, cin, cout, k, s, relu)


def conv_transpose2d(x, pct, out=None, c_off=0):
    n, h, w, c = x.shape
    s = pct.stride
    c_out = pct.subs[0][0].cout
    if out is None:
        out = torch.empty((n, h * s, w * s, c_out), dtype=torch.bfloat16, device=x.device)
    for pc, m in pct.subs:
        attrs = _ConvAttrs(pc.kh, pc.kw, 1, 0, int(pct.relu), int(CONV_VARIANT))
        attrs.adv, attrs.pad_top, attrs.pad_left, attrs.sub_h, attrs.sub_w = 1, m["pad_top"], m["pad_left"], h, w
        attrs.out_stride, attrs.out_off_y, attrs.out_off_x, attrs.c_off, attrs.cout = s, m["py"], m["px"], int(c_off), pc.cout
        attrs.korder = getattr(pc, "korder", 0)
        attrs.tune = TUNE
        _lib.call("md_conv2d", [x, pc.w, pc.bias, None, out], extra=attrs)
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


class _SppfAttrs(ctypes.Structure):
    _fields_ = [("channels", ctypes.c_int32), ("k", ctypes.c_int32)]


def sppf_pool_fits(h, w, c):
    """True when md_sppf_pool can hold an h x w image's working set in LDS (else: three maxpool2d launches)"""
    return _lib.lib().md_sppf_pool_groups(int(h), int(w), int(c)) > 0


def sppf_pool(buf, c, k=5):
    """SPPF's pooling chain in place on its concat buffer [N,H,W,>=4c]: channels [c,2c) = mp(x), [2c,3c) = mp(mp(x)), [3c,4c) = mp(mp(mp(x)))
    with x = channels [0,c) and mp = max-pool k x k / stride 1 / pad k//2 (torch semantics), one md_sppf_pool launch."""
    _lib.call("md_sppf_pool", [buf], extra=_SppfAttrs(int(c), int(k)))
    return buf


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


def nhwc_to_nchw_f32(x, c0, width):
    """x [N,H,W,C] bf16 -> [N,width,H,W] fp32 (channels c0..c0+width)."""
    n, h, w, _ = x.shape
    y = torch.empty((n, width, h, w), dtype=torch.float32, device=x.device)
    _lib.call("md_nhwc_to_nchw_f32", [x, y], extra=_SliceAttrs(c0, width))
    return y


def concat_copy(src, dst, c0):
    """dst[..., c0:c0+C] = src (channel concat of a tensor that could not be produced in place)."""
    _lib.call("md_concat_copy", [src, dst], extra=_SliceAttrs(int(c0), src.shape[3]))
    return dst


class _Upsample2xAttrs(ctypes.Structure):
    _fields_ = [("c0", ctypes.c_int32), ("width", ctypes.c_int32), ("src_c0", ctypes.c_int32)]


def upsample2x(src, dst=None, c0=0, src_c0=0, width=None):
    """Nearest 2x upsample of channels [src_c0, src_c0 + width) of `src` (default: all of it), optionally straight into channels
    [c0, c0 + width) of a wider `dst`."""
    n, h, w, cs = src.shape
    c = cs - src_c0 if width is None else width
    if dst is None:
        dst = torch.empty((n, 2 * h, 2 * w, c), dtype=torch.bfloat16, device=src.device)
    _lib.call("md_upsample2x", [src, dst], extra=_Upsample2xAttrs(int(c0), int(c), int(src_c0)))
    return dst


# ----------------------------------------------------------------------------- fused ResNet stem (csrc/stem.hip)
STEM_PAD_LO, STEM_PAD_HI = 7, 9


class PackedStem:
    def __init__(self, w, bias):
        self.w, self.bias = w, bias

    def to(self, device):
        self.w, self.bias = self.w.to(device), self.bias.to(device)
        return self


def pack_stem(weight, bn=None, bias=None):
    """weight [64,3,7,7] fp32 (+ eval BatchNorm) -> md_stem_pool operands: w [64, 224] bf16 with K = (ky, kx 0..7, c 0..3)."""
    weight = weight.detach().to(torch.float32)
    cout, cin, kh, kw = weight.shape
    if (cout, kh, kw) != (64, 7, 7) or cin > 3:
        raise _lib.MindDetHipError("pack_stem: expects a [64, <=3, 7, 7] stem convolution")
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        scale = gamma.to(torch.float32) / torch.sqrt(var.to(torch.float32) + eps)
        weight = weight * scale.view(-1, 1, 1, 1)
        b = beta.to(torch.float32) - mean.to(torch.float32) * scale
        if bias is not None:
            b = b + bias.to(torch.float32) * scale
    else:
        b = bias.detach().to(torch.float32) if bias is not None else torch.zeros(cout)
    wp = torch.zeros((64, 7, 8, 4), dtype=torch.float32)
    wp[:, :, :7, :cin] = weight.permute(0, 2, 3, 1)
    return PackedStem(wp.reshape(64, 224).to(torch.bfloat16).contiguous(), b.contiguous())


class _StemConvAttrs(ctypes.Structure):
    _fields_ = [("kh", ctypes.c_int32), ("act", ctypes.c_int32)]


class PackedStemConv:
    def __init__(self, w, bias, kh, act, cout):
        self.w, self.bias, self.kh, self.act, self.cout = w, bias, kh, act, cout

    def to(self, device):
        self.w, self.bias = self.w.to(device), self.bias.to(device)
        return self


def pack_stem_conv(weight, bn=None, bias=None, act=None):
    """weight [Cout, <=3, k, k] fp32 (+ eval BatchNorm) of a stride-2 stem conv with k = 6 (pad 2) or k = 3 (pad 1), Cout 32 or 64
    -> md_stem_conv operands, or None when the layer is not one it takes.  K = (ky, kx', c 0..3): k = 6 -> kx' = kx + 1 in 0..7,
    k = 3 -> kx' = kx in 0..3."""
    weight = weight.detach().to(torch.float32)
    cout, cin, kh, kw = weight.shape
    if kh != kw or kh not in (6, 3) or cin > 3 or cout not in (32, 64):
        return None
    if bn is not None:
        gamma, beta, mean, var, eps = bn
        scale = gamma.to(torch.float32) / torch.sqrt(var.to(torch.float32) + eps)
        weight = weight * scale.view(-1, 1, 1, 1)
        b = beta.to(torch.float32) - mean.to(torch.float32) * scale
        if bias is not None:
            b = b + bias.to(torch.float32) * scale
    else:
        b = bias.detach().to(torch.float32) if bias is not None else torch.zeros(cout)
    kx, x0 = (8, 1) if kh == 6 else (4, 0)
    wp = torch.zeros((cout, kh, kx, 4), dtype=torch.float32)
    wp[:, :, x0:x0 + kw, :cin] = weight.permute(0, 2, 3, 1)
    code = {None: 0, False: 0, True: 1, "relu": 1, "silu": 2, 0: 0, 1: 1, 2: 2}[act]
    return PackedStemConv(wp.reshape(cout, kh * kx * 4).to(torch.bfloat16).contiguous(), b.contiguous(), kh, code, cout)


def stem_conv(x4, ps):
    """x4: the batch in the stem layout [N, H+16, W+16, 4] -> [N, H/2, W/2, Cout] bf16 (md_stem_conv)."""
    n, hp, wp, c = x4.shape
    h, w = hp - STEM_PAD_LO - STEM_PAD_HI, wp - STEM_PAD_LO - STEM_PAD_HI
    y = torch.empty((n, h // 2, w // 2, ps.cout), dtype=torch.bfloat16, device=x4.device)
    _lib.call("md_stem_conv", [x4, ps.w, ps.bias, y], extra=_StemConvAttrs(ps.kh, ps.act))
    return y


def stem_layout_ok(h, w):
    return h % 16 == 0 and w % 64 == 0


def to_stem_layout(x):
    """[N,H,W,>=3] bf16 NHWC image batch -> [N, H+16, W+16, 4] (the input layout of md_stem_pool).  Layout plumbing for
    callers that hold the 8-channel format; a pre-processing pipeline writes this layout directly."""
    n, h, w, _ = x.shape
    out = torch.zeros((n, h + STEM_PAD_LO + STEM_PAD_HI, w + STEM_PAD_LO + STEM_PAD_HI, 4), dtype=torch.bfloat16, device=x.device)
    out[:, STEM_PAD_LO:STEM_PAD_LO + h, STEM_PAD_LO:STEM_PAD_LO + w, :3] = x[..., :3]
    return out


def stem_pool(x4, ps):
    n, hp, wp, c = x4.shape
    h, w = hp - STEM_PAD_LO - STEM_PAD_HI, wp - STEM_PAD_LO - STEM_PAD_HI
    y = torch.empty((n, h // 4, w // 4, 64), dtype=torch.bfloat16, device=x4.device)
    _lib.call("md_stem_pool", [x4, ps.w, ps.bias, y])
    return y


# ----------------------------------------------------------------------------- DCNv2 (csrc/dcn.hip)
class _PoolAttrs3(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int32), ("stride", ctypes.c_int32), ("pad", ctypes.c_int32), ("zero_pad", ctypes.c_int32)]


def deform_conv2d(x, pc_offset, pc, relu=None):
    """ModulatedDeformConv2d (centernet/src/resnet.py:24-106): offset conv (md_conv2d, 3*k*k channels) -> deformable
    im2col (md_deform_cols) -> the main conv as a 1x1 GEMM over the k*k*C columns (md_conv2d).  `pc` is the layer's packed
    k x k conv with K order (tap, channel) (pack_conv(..., korder=0))."""
    n, h, w, c = x.shape
    if getattr(pc, "korder", 0) != 0 or pc.kh != pc.kw or pc.cin != c:
        raise _lib.MindDetHipError("deform_conv2d: the main conv must be packed with korder=0 for this input")
    off = conv2d(x, pc_offset)                                   # [N,Ho,Wo,>=3*k*k]
    ho, wo = off.shape[1], off.shape[2]
    cols = torch.empty((n, ho, wo, pc.kh * pc.kw * c), dtype=torch.bfloat16, device=x.device)
    _lib.call("md_deform_cols", [x, off, cols], extra=_PoolAttrs3(pc.kh, pc.stride, pc.pad, 0))
    pw = PackedConv(pc.w, pc.bias, pc.kh * pc.kw * c, pc.cout, 1, 1, 1, 0, pc.relu)
    pw.cin_real, pw.korder = pc.kh * pc.kw * c, 0
    return conv2d(cols, pw, relu=relu)


# ----------------------------------------------------------------------------- image pre-processing (csrc/preproc.hip)
class _PreAttrs(ctypes.Structure):
    _fields_ = [("out_h", ctypes.c_int32), ("out_w", ctypes.c_int32), ("pad_lo", ctypes.c_int32), ("pad_hi", ctypes.c_int32)]


def image_preprocess(img_u8, mat, mean, std, out_hw, stem_layout=True):
    """uint8 [N,Hs,Ws,3] device images -> normalised bf16 network input, warped by `mat` [N,6] (output pixel -> source pixel;
    e.g. the inverse of get_affine_transform(c, s, 0, (out_w, out_h)) of centernet/src/image.py:25-57).  stem_layout: the
    zero-bordered 4-channel layout md_stem_pool consumes, else [N,out_h,out_w,8]."""
    n = img_u8.shape[0]
    ho, wo = out_hw
    lo, hi, c = (STEM_PAD_LO, STEM_PAD_HI, 4) if stem_layout else (0, 0, 8)
    out = torch.empty((n, ho + lo + hi, wo + lo + hi, c), dtype=torch.bfloat16, device=img_u8.device)
    norm = torch.tensor(list(mean) + list(std), dtype=torch.float32, device=img_u8.device)
    _lib.call("md_image_preprocess", [img_u8, mat, norm, out], extra=_PreAttrs(ho, wo, lo, hi))
    return out
