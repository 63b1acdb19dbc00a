"""ctypes loader for libminddet_hip.so and the generic AOT-ABI call helper.

The call convention is the reference's MindSpore AOT custom-op ABI
(minddet/models/centerpoint/det3d_ms/ops/test_custom_pytorch/iou3d_nms_kernel.cu:445):
``int op(int nparam, void** params, int* ndims, int64_t** shapes, const char** dtypes,
void* stream, void* extra)``; see include/minddet_hip.h.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# MD_DIAG_LIB=1 (tools/pp_stamps.py, tools/igemm_stamps.py only): the -DMD_DIAG build with the timing ablations and stamp kernels
LIB_PATH = os.path.join(_HERE, "libminddet_hip_diag.so" if os.environ.get("MD_DIAG_LIB") == "1" else "libminddet_hip.so")
_lib = None

_DT = {
    "torch.float32": b"float32", "torch.float16": b"float16", "torch.bfloat16": b"bfloat16",
    "torch.float64": b"float64", "torch.int32": b"int32", "torch.int64": b"int64",
    "torch.uint8": b"uint8", "torch.int8": b"int8", "torch.int16": b"int16", "torch.bool": b"uint8",
}

ERRORS = {1: "wrong nparam", 2: "bad dtype/shape/argument", 3: "HIP runtime error", 4: "unsupported size"}


class MindDetHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile every HIP source for gfx950 into minddet_amd/libminddet_hip.so (in-tree)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return LIB_PATH


def lib():
    """Load the library; raise loudly if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MindDetHipError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C minddet_amd/csrc`. minddet_amd has no CPU/PyTorch fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.md_version.restype = ctypes.c_char_p
    return _lib


def exported_symbols():
    """Names declared in include/minddet_hip.h (parsed), used by the CPU-side ABI test."""
    import re

    hdr = os.path.join(os.path.dirname(_HERE), "include", "minddet_hip.h")
    txt = open(hdr).read()
    names = re.findall(r"^\s*int\s+(\w+)\s*\(MD_AOT_ARGS\)\s*;", txt, flags=re.M)
    names += re.findall(r"^\s*const char \*(\w+)\s*\(void\)\s*;", txt, flags=re.M)
    return names


def call(name, tensors, extra=None, stream=None):
    """Invoke AOT op `name`. `tensors`: list of torch CUDA tensors (or None for a NULL param)
    in the op's documented order (inputs, outputs[, workspace]). `extra`: ctypes struct or None."""
    import torch

    fn = getattr(lib(), name)
    n = len(tensors)
    params = (ctypes.c_void_p * n)()
    ndims = (ctypes.c_int * n)()
    shape_bufs = []
    shapes = (ctypes.POINTER(ctypes.c_int64) * n)()
    dtypes = (ctypes.c_char_p * n)()
    dev = None
    for i, t in enumerate(tensors):
        if t is None:
            params[i] = None
            ndims[i] = 0
            buf = (ctypes.c_int64 * 1)(0)
            dtypes[i] = None
        else:
            if not t.is_cuda:
                raise MindDetHipError(f"{name}: param {i} is not a device tensor")
            if not t.is_contiguous():
                raise MindDetHipError(f"{name}: param {i} is not contiguous")
            dev = t.device if dev is None else dev
            params[i] = t.data_ptr()
            ndims[i] = t.dim()
            buf = (ctypes.c_int64 * max(t.dim(), 1))(*t.shape)
            dtypes[i] = _DT[str(t.dtype)]
        shape_bufs.append(buf)
        shapes[i] = ctypes.cast(buf, ctypes.POINTER(ctypes.c_int64))
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    ex = ctypes.byref(extra) if extra is not None else None
    with torch.cuda.device(dev):
        rc = fn(ctypes.c_int(n), params, ndims, shapes, dtypes, ctypes.c_void_p(stream), ex)
    if rc != 0:
        raise MindDetHipError(f"{name} failed: rc={rc} ({ERRORS.get(rc, 'unknown')})")
    return rc
