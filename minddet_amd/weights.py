"""Weight import / export for the reference's checkpoints.

Formats
  * MindSpore ``.ckpt``: a protobuf ``Checkpoint { repeated Value value = 1 }`` with ``Value { string tag = 1; TensorProto tensor = 2 }``
    and ``TensorProto { repeated int64 dims = 1; string tensor_type = 2; bytes tensor_content = 3 }`` (MindSpore
    ``mindspore/ccsrc/utils/checkpoint.proto``; what ``save_checkpoint`` writes and centernet/convert_ckpt.py produces).  Parsed
    and written here with a 40-line wire-format codec: no MindSpore, no protobuf package, nothing executed from the file.
  * torch ``.pth`` state dicts: ``torch.load(..., weights_only=True)``.

Key maps (CenterNet, the model the reference ships converters and key lists for):
  ``centernet/centernet_ms_params.txt`` (151 MindSpore keys: ``network.backbone.*``, ``network.deconv_layers.N.*``,
  ``network.{hm,wh,reg}_fn.{0,2}.*``) and ``centernet/centernet_params.txt`` (the 151 torch keys of the original CenterNet
  release, same order).  centernet/convert_ckpt.py:56-92 pairs the two lists POSITIONALLY and then swaps the BatchNorm names
  (moving_mean <-> gamma, moving_variance <-> beta), because torch lists BN as (weight, bias, running_mean, running_var) and
  MindSpore as (moving_mean, moving_variance, gamma, beta); the net effect is the natural map weight -> gamma, bias -> beta,
  running_mean -> moving_mean, running_var -> moving_variance, which is what `torch_to_ms_name` implements directly.

`load_centernet` writes the tensors into the fp32 reference-layout parameters of `graphs.CenterNet` (``.weight [Cout,Cin,kh,kw]``,
``.bn = (gamma, beta, mean, var, eps)``, ...); `model.to(device)` then folds BN and packs them for the kernels as usual.
No checkpoint ships with the reference and there is no network here: the tests round-trip synthetic checkpoints through
both formats and both namings (parity of real weights: unpinned).
"""
import re
import struct

import numpy as np
import torch

_MS_DTYPES = {"Float32": np.float32, "Float16": np.float16, "Float64": np.float64, "Int32": np.int32, "Int64": np.int64,
              "Int8": np.int8, "UInt8": np.uint8, "Bool": np.bool_}
_NP_TO_MS = {np.dtype(v).str: k for k, v in _MS_DTYPES.items()}


# ----------------------------------------------------------------------------- protobuf wire format (subset)
def _varint(buf, i):
    r, shift = 0, 0
    while True:
        b = buf[i]
        i += 1
        r |= (b & 0x7F) << shift
        if not b & 0x80:
            return r, i
        shift += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one message; value = int (varint / fixed) or memoryview (length-delimited)."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 2:
            ln, i = _varint(buf, i)
            v = buf[i:i + ln]
            i += ln
        elif wt == 1:
            v = struct.unpack_from("<q", buf, i)[0]
            i += 8
        elif wt == 5:
            v = struct.unpack_from("<i", buf, i)[0]
            i += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _enc_varint(v):
    out = bytearray()
    v &= (1 << 64) - 1
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _enc_field(fno, payload):
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def read_ms_ckpt(path):
    """MindSpore checkpoint -> {name: np.ndarray}.  Only tag / dims / tensor_type / tensor_content are read."""
    buf = memoryview(open(path, "rb").read())
    out = {}
    for fno, wt, val in _fields(buf):
        if fno != 1 or wt != 2:
            continue
        tag, tensor = None, None
        for f2, w2, v2 in _fields(val):
            if f2 == 1 and w2 == 2:
                tag = bytes(v2).decode("utf-8")
            elif f2 == 2 and w2 == 2:
                tensor = v2
        if tag is None or tensor is None:
            continue
        dims, ttype, content = [], None, b""
        for f3, w3, v3 in _fields(tensor):
            if f3 == 1 and w3 == 0:
                dims.append(v3)
            elif f3 == 1 and w3 == 2:  # packed repeated int64
                j = 0
                while j < len(v3):
                    d, j = _varint(v3, j)
                    dims.append(d)
            elif f3 == 2 and w3 == 2:
                ttype = bytes(v3).decode("utf-8")
            elif f3 == 3 and w3 == 2:
                content = bytes(v3)
        if ttype not in _MS_DTYPES:
            raise ValueError(f"{tag}: unsupported tensor_type {ttype!r}")
        arr = np.frombuffer(content, dtype=_MS_DTYPES[ttype]).copy()
        out[tag] = arr.reshape(dims) if dims else arr.reshape(())
    return out


def write_ms_ckpt(path, params):
    """{name: array-like} -> MindSpore checkpoint bytes (the same three fields save_checkpoint writes)."""
    with open(path, "wb") as f:
        for name, value in params.items():
            a = np.ascontiguousarray(value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value))
            if a.dtype.str not in _NP_TO_MS:
                raise ValueError(f"{name}: dtype {a.dtype} has no MindSpore tensor_type here")
            tensor = b"".join(_enc_varint((1 << 3) | 0) + _enc_varint(int(d)) for d in a.shape)
            tensor += _enc_field(2, _NP_TO_MS[a.dtype.str].encode()) + _enc_field(3, a.tobytes())
            f.write(_enc_field(1, _enc_field(1, name.encode()) + _enc_field(2, tensor)))


def read_torch_pth(path):
    sd = torch.load(path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return {k[len("module."):] if k.startswith("module.") else k: v.numpy() for k, v in sd.items() if isinstance(v, torch.Tensor)}


# ----------------------------------------------------------------------------- CenterNet key map
_BN_T2M = {"weight": "gamma", "bias": "beta", "running_mean": "moving_mean", "running_var": "moving_variance"}


def torch_to_ms_name(name):
    """Original-CenterNet torch key -> the reference's MindSpore key (net effect of centernet/convert_ckpt.py:56-92)."""
    if name.endswith("num_batches_tracked"):
        return None
    head = re.match(r"^(hm|wh|reg)\.(\d)\.(weight|bias)$", name)
    if head:
        return f"network.{head.group(1)}_fn.{head.group(2)}.{head.group(3)}"
    name = name.replace("conv_offset_mask", "conv_offset")
    prefix = "network." if name.startswith("deconv_layers.") else "network.backbone."
    parts = name.split(".")
    leaf = parts[-1]
    is_bn = (re.search(r"(^|\.)bn\d\.", name) is not None or re.search(r"downsample\.1\.", name) is not None or
             (parts[0] == "deconv_layers" and int(parts[1]) % 3 == 1 and "conv_offset" not in name))
    if is_bn and leaf in _BN_T2M:
        parts[-1] = _BN_T2M[leaf]
    return prefix + ".".join(parts)


def centernet_state(model, naming="ms"):
    """The model's parameters under the reference's names (naming 'ms': centernet_ms_params.txt; 'torch': centernet_params.txt)."""
    out = {}

    def bn(prefix, m):
        gamma, beta, mean, var, _eps = m.bn
        if naming == "ms":
            out[prefix + ".moving_mean"], out[prefix + ".moving_variance"] = mean, var
            out[prefix + ".gamma"], out[prefix + ".beta"] = gamma, beta
        else:
            out[prefix + ".weight"], out[prefix + ".bias"] = gamma, beta
            out[prefix + ".running_mean"], out[prefix + ".running_var"] = mean, var

    root = "network." if naming == "ms" else ""
    bb = root + ("backbone." if naming == "ms" else "")
    out[bb + "conv1.weight"] = model.backbone.conv1.weight
    bn(bb + "bn1", model.backbone.conv1)
    for li, stage in enumerate(model.backbone.stages, 1):
        for bi, blk in enumerate(stage):
            p = f"{bb}layer{li}.{bi}."
            for ci, m in enumerate([m for m in blk.modules() if m is not blk.downsample], 1):
                out[p + f"conv{ci}.weight"] = m.weight
                bn(p + f"bn{ci}", m)
            if blk.downsample is not None:
                out[p + "downsample.0.weight"] = blk.downsample.weight
                bn(p + "downsample.1", blk.downsample)
    off = "conv_offset" if naming == "ms" else "conv_offset_mask"
    for i in range(len(model.neck) // 2):
        dcn, dec = model.neck[2 * i], model.neck[2 * i + 1]
        p = f"{root}deconv_layers.{6 * i}"
        out[p + ".weight"] = dcn.weight
        out[p + ".bias"] = dcn.bias if dcn.bias is not None else torch.zeros(dcn.cout)
        if hasattr(dcn, "offset_weight"):
            out[f"{p}.{off}.weight"], out[f"{p}.{off}.bias"] = dcn.offset_weight, dcn.offset_bias
        bn(f"{root}deconv_layers.{6 * i + 1}", dcn)
        out[f"{root}deconv_layers.{6 * i + 3}.weight"] = dec.weight_t
        bn(f"{root}deconv_layers.{6 * i + 4}", dec)
    for name in ("hm", "wh", "reg"):
        c1, c2 = model.heads[name]
        p = f"{root}{name}_fn" if naming == "ms" else name
        out[p + ".0.weight"], out[p + ".0.bias"] = c1.weight, c1.bias
        out[p + ".2.weight"], out[p + ".2.bias"] = c2.weight, c2.bias
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}


def load_centernet(model, params, naming="auto", strict=True):
    """Write a {name: array} checkpoint (MindSpore or torch naming) into a graphs.CenterNet; returns the unused keys.
    Call model.to(device) afterwards to fold / pack."""
    if naming == "auto":
        naming = "ms" if any(k.startswith("network.") for k in params) else "torch"
    if naming == "torch":
        params = {torch_to_ms_name(k): v for k, v in params.items() if torch_to_ms_name(k) is not None}
    want = centernet_state(model, "ms")
    used = set()

    def get(key, like):
        if key not in params:
            if strict:
                raise KeyError(f"checkpoint has no {key!r}")
            return None
        a = np.asarray(params[key])
        if tuple(a.shape) != tuple(like.shape):
            raise ValueError(f"{key}: checkpoint shape {tuple(a.shape)} != model shape {tuple(like.shape)}")
        used.add(key)
        return torch.from_numpy(a.astype(np.float32))

    def set_bn(prefix, m):
        vals = [get(prefix + s, want[prefix + s]) for s in (".gamma", ".beta", ".moving_mean", ".moving_variance")]
        if all(v is not None for v in vals):
            m.bn = (vals[0], vals[1], vals[2], vals[3], m.bn[4])

    def set_attr(m, attr, key):
        v = get(key, want[key])
        if v is not None:
            setattr(m, attr, v)

    bb = "network.backbone."
    set_attr(model.backbone.conv1, "weight", bb + "conv1.weight")
    set_bn(bb + "bn1", model.backbone.conv1)
    for li, stage in enumerate(model.backbone.stages, 1):
        for bi, blk in enumerate(stage):
            p = f"{bb}layer{li}.{bi}."
            for ci, m in enumerate([m for m in blk.modules() if m is not blk.downsample], 1):
                set_attr(m, "weight", p + f"conv{ci}.weight")
                set_bn(p + f"bn{ci}", m)
            if blk.downsample is not None:
                set_attr(blk.downsample, "weight", p + "downsample.0.weight")
                set_bn(p + "downsample.1", blk.downsample)
    for i in range(len(model.neck) // 2):
        dcn, dec = model.neck[2 * i], model.neck[2 * i + 1]
        p = f"network.deconv_layers.{6 * i}"
        set_attr(dcn, "weight", p + ".weight")
        set_attr(dcn, "bias", p + ".bias")
        if hasattr(dcn, "offset_weight"):
            set_attr(dcn, "offset_weight", p + ".conv_offset.weight")
            set_attr(dcn, "offset_bias", p + ".conv_offset.bias")
        set_bn(f"network.deconv_layers.{6 * i + 1}", dcn)
        set_attr(dec, "weight_t", f"network.deconv_layers.{6 * i + 3}.weight")
        set_bn(f"network.deconv_layers.{6 * i + 4}", dec)
    for name in ("hm", "wh", "reg"):
        c1, c2 = model.heads[name]
        set_attr(c1, "weight", f"network.{name}_fn.0.weight")
        set_attr(c1, "bias", f"network.{name}_fn.0.bias")
        set_attr(c2, "weight", f"network.{name}_fn.2.weight")
        set_attr(c2, "bias", f"network.{name}_fn.2.bias")
    model.fuse_heads()
    return sorted(k for k in params if k not in used)


# ----------------------------------------------------------------------------- CenterPoint / PointPillars
def torch_to_ms_generic(names):
    """det3d (torch) state-dict keys -> the reference's MindSpore keys, the rule of CenterPoint's `convert()`
    (centerpoint/det3d_ms/models/detectors/point_pillars.py:137-168): `num_batches_tracked` / `global_step` dropped;
    `running_mean` -> `moving_mean`, `running_var` -> `moving_variance`; `weight` -> `gamma` and `bias` -> `beta` ONLY for
    BatchNorm layers, recognised by a sibling `running_var` key (a conv / dense `weight` or `bias` keeps its name).
    Returns {torch key: MindSpore key} for the kept keys."""
    keys = set(names)
    out = {}
    for k in sorted(keys):
        if "num_batches_tracked" in k or "global_step" in k:
            continue
        if "running_mean" in k:
            out[k] = k.replace("running_mean", "moving_mean")
        elif "running_var" in k:
            out[k] = k.replace("running_var", "moving_variance")
        elif "bias" in k:
            out[k] = k.replace("bias", "beta") if k.replace("bias", "running_var") in keys else k
        elif "weight" in k:
            out[k] = k.replace("weight", "gamma") if k.replace("weight", "running_var") in keys else k
        else:
            out[k] = k
    return out


def strip_net_prefix(params):
    """PointPillars' `get_params_for_net` (pointpillars/src/utils.py:48-56): a training checkpoint stores the network under
    `network.network.` (TrainOneStepCell(WithLossCell(net))) and optimizer copies under `optimizer.`; both prefixes are
    stripped, every other key is dropped."""
    out = {}
    for k, v in params.items():
        if k.startswith("optimizer."):
            out[k[10:]] = v
        elif k.startswith("network.network."):
            out[k[16:]] = v
    return out


def _rpn_slots(model, prefix):
    """(module, key prefix of its conv, key prefix of its BN) for every layer of graphs.RPN under the reference's cell names
    (centerpoint/det3d_ms/models/necks/rpn.py:114-143: block i = SequentialCell[Pad, Conv, BN, ReLU, (Conv, BN, ReLU) x n] ->
    conv at 1 + 3j, BN at 2 + 3j; :60-105: deblock k = SequentialCell[Conv2dTranspose | Conv2d, BN, ReLU] -> 0 and 1)."""
    slots = []
    for i, blk in enumerate(model.blocks):
        for j, m in enumerate(blk):
            slots.append((m, f"{prefix}blocks.{i}.{1 + 3 * j}", f"{prefix}blocks.{i}.{2 + 3 * j}"))
    for k, m in enumerate(model.deblocks):
        slots.append((m, f"{prefix}deblocks.{k}.0", f"{prefix}deblocks.{k}.1"))
    return slots


def rpn_state(model, prefix="neck.", naming="ms"):
    """graphs.RPN parameters under the reference's names (MindSpore, or the det3d torch original with naming='torch')."""
    bn_names = ("gamma", "beta", "moving_mean", "moving_variance") if naming == "ms" else ("weight", "bias", "running_mean", "running_var")
    out = {}
    for m, conv, bn in _rpn_slots(model, prefix):
        out[conv + ".weight"] = m.weight_t if hasattr(m, "weight_t") else m.weight
        for n, v in zip(bn_names, m.bn[:4]):
            out[f"{bn}.{n}"] = v
    return {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()}


def load_rpn(model, params, prefix="neck.", naming="auto", strict=True):
    """Write a CenterPoint / PointPillars checkpoint's neck into a graphs.RPN; returns the unused keys.  naming 'torch' goes
    through `torch_to_ms_generic` first; call model.to(device) afterwards to fold / pack."""
    if naming == "auto":
        naming = "torch" if any(k.endswith("running_var") for k in params) else "ms"
    if naming == "torch":
        kmap = torch_to_ms_generic(params.keys())
        params = {kmap[k]: v for k, v in params.items() if k in kmap}
    used = set()

    def get(key, like):
        if key not in params:
            if strict:
                raise KeyError(f"checkpoint has no {key!r}")
            return None
        a = np.asarray(params[key])
        if tuple(a.shape) != tuple(like.shape):
            raise ValueError(f"{key}: checkpoint shape {tuple(a.shape)} != model shape {tuple(like.shape)}")
        used.add(key)
        return torch.from_numpy(a.astype(np.float32))

    for m, conv, bn in _rpn_slots(model, prefix):
        attr = "weight_t" if hasattr(m, "weight_t") else "weight"
        w = get(conv + ".weight", getattr(m, attr))
        if w is not None:
            setattr(m, attr, w)
        vals = [get(f"{bn}.{n}", m.bn[i]) for i, n in enumerate(("gamma", "beta", "moving_mean", "moving_variance"))]
        if all(v is not None for v in vals):
            m.bn = (vals[0], vals[1], vals[2], vals[3], m.bn[4])
    return sorted(k for k in params if k not in used)
