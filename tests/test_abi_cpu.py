"""CPU-side checks of the drop-in boundary: the library loads and exports every symbol
include/minddet_hip.h declares (no compute without a GPU), and the product package never
touches oracle/."""
import ctypes
import os
import re

import pytest

from minddet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = _lib.exported_symbols()
    assert len(names) >= 9
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/minddet_hip.h but not exported"
    lib.md_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.md_version()


def test_reference_symbol_names_are_kept():
    # the names the reference's ops.Custom strings bind (iou_gpu.py:18,33,53,72)
    for n in ["BoxesIouBevGpu", "BoxesOverlapBevGpu", "NmsGpu", "NmsNormalGpu"]:
        assert n in _lib.exported_symbols()


def test_wrong_nparam_is_an_error_code_not_a_crash():
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in ["NmsGpu", "boxes_iou_nms_gpu", "BoxesIouBevGpu", "md_nms_aligned"]:
        rc = getattr(lib, n)(1, None, None, None, None, None, None)
        assert rc == 1  # iou-bev-nms-org.cpp:238 convention


def test_round4_ops_argument_checks_without_a_gpu():
    """md_c3_pair / md_sppf_pool refuse a wrong parameter count before touching anything, and md_sppf_pool_groups (a pure function: how
    many 8-channel groups one workgroup holds in LDS, 0 = the map does not fit and the caller keeps three md_maxpool2d launches) answers on the CPU"""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in ["md_c3_pair", "md_sppf_pool", "md_topk_segmented", "md_upsample2x"]:
        assert getattr(lib, n)(1 if n != "md_sppf_pool" else 3, None, None, None, None, None, None) in (1, 2)
    g = lib.md_sppf_pool_groups
    assert g(20, 20, 256) == 4 and g(20, 20, 8) == 1 and g(40, 40, 128) == 1   # 400 px x 4 groups x 64 B = 100 KiB; 1600 px x 1 x 64 B = 100 KiB
    assert g(80, 80, 64) == 0 and g(0, 20, 64) == 0 and g(20, 20, 12) == 0      # too large for LDS / empty / channels not a multiple of 8


def test_no_process_wide_tuning_setters_and_reserved0_is_checked():
    """include/minddet_hip.h promises 're-entrant, no global mutable state': the conv family's tuning knobs are per-call attributes
    (md_conv_tune), the library exports no setter, and md_conv2d_attrs.reserved0 != 0 is MD_ERR_ARG (checked before any device call,
    so this runs without a GPU)."""
    lib = ctypes.CDLL(_lib.LIB_PATH)
    hdr = open(os.path.join(ROOT, "include", "minddet_hip.h")).read()
    assert "md_conv2d_set_" not in hdr
    for n in ("md_conv2d_set_chunk_limit", "md_conv2d_set_stream_rounds", "md_conv2d_set_stream_tune", "md_conv2d_set_pers_min_k",
              "md_conv2d_set_dual_pp_min_k"):
        assert not hasattr(lib, n), f"{n}: a process-wide knob is back in the product library"
    from minddet_amd import nn_ops

    # the ctypes mirror has the header's field list (23 int32 + 6 int32 of md_conv_tune)
    m = re.search(r"typedef struct md_conv2d_attrs \{(.*?)\} md_conv2d_attrs;", hdr, flags=re.S)
    body = re.sub(r"/\*.*?\*/", "", m.group(1), flags=re.S)
    n_i32 = sum(len(d.split(",")) for d in re.findall(r"int32_t\s+([^;]+);", body))
    assert n_i32 == 23 and "md_conv_tune tune;" in body
    assert ctypes.sizeof(nn_ops._ConvAttrs) == 4 * (23 + 6) and ctypes.sizeof(nn_ops.ConvTune) == 24
    assert ctypes.sizeof(nn_ops._DualAttrs) == 4 * (2 + 6)
    # reserved0 != 0 -> rc 2; dummy host pointers are never dereferenced on this path
    n = 5
    dummy = (ctypes.c_char * 64)()
    params = (ctypes.c_void_p * n)(*[ctypes.addressof(dummy)] * n)
    ndims = (ctypes.c_int * n)(4, 2, 1, 4, 4)
    sh = [(ctypes.c_int64 * 4)(1, 8, 8, 64), (ctypes.c_int64 * 2)(64, 64), (ctypes.c_int64 * 1)(64), (ctypes.c_int64 * 4)(1, 8, 8, 64),
          (ctypes.c_int64 * 4)(1, 8, 8, 64)]
    shapes = (ctypes.POINTER(ctypes.c_int64) * n)(*[ctypes.cast(b, ctypes.POINTER(ctypes.c_int64)) for b in sh])
    dtypes = (ctypes.c_char_p * n)(b"bfloat16", b"bfloat16", b"float32", b"bfloat16", b"bfloat16")
    attrs = nn_ops._ConvAttrs(1, 1, 1, 0, 0, 0)
    attrs.reserved0 = 1
    assert lib.md_conv2d(n, params, ndims, shapes, dtypes, None, ctypes.byref(attrs)) == 2


def test_product_never_imports_oracle():
    pat = re.compile(r"^\s*(from|import)\s+oracle\b|oracle[./]_ref|liboracle", re.M)
    for d, _, files in os.walk(os.path.join(ROOT, "minddet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(d, f)).read()
                assert not pat.search(txt), f"{f} references the oracle"
    for sub in ("minddet", "tools", "configs"):
        for d, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith((".py", ".sh")):
                    assert not pat.search(open(os.path.join(d, f)).read()), f"{sub}/{f} references the oracle"
    # bench.py may use it in its cpu_baseline leg only, __graft_entry__.py in build() / smoke() only
    bench = open(os.path.join(ROOT, "bench.py")).read()
    import ast

    tree = ast.parse(bench)
    users = set()
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for n in ast.walk(fn):
            if (isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names)) or \
                    (isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle"):
                users.add(fn.name)
    # reference_ops_baseline IS the cpu_baseline.reference_ops leg; in main() the import sits behind `cpu_baseline = None`
    assert users <= {"main", "reference_ops_baseline"}, users
    main_src = bench[bench.index("def main("):]
    assert main_src.index("cpu_baseline = None") < pat.search(main_src).start()
    assert not any(isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n) for n in tree.body)   # never at module level


def test_committed_bench_line_has_the_contract_fields():
    """profiles/r03_bench.json is the bench.py line of the profiled run: the driver's contract fields, the roofline object and the
    cpu_baseline object must all be there (a schema regression in bench.py shows up when the profile is regenerated)."""
    import json

    line = open(os.path.join(ROOT, "profiles", "r03_bench.json")).read().strip().splitlines()[-1]
    j = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["dtype"] == "bf16"
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("reference", "port")
    # round 2: the reference's own compiled operator timed beside the device ops, the zero-operand replay, the uint8 start
    ro = c["reference_ops"]
    assert ro["kind"] == "reference" and ro["boxes_iou_nms"]["keep_lists_identical"] is True and ro["boxes_iou_bev"]["cpu_ms"] > 0
    assert r["zero_operands"]["zero_frac"] >= r["zero_operands"]["random_frac"] and j["from_uint8"]["ms_per_step"] > 0
    assert j["nms_prefix"]["image_slots_flagged"] == 0 and j["n_gpus"] == 1
    # round 3: the host's enqueue cost beside the step time (multi-GPU readiness), PMC traffic present only because the file's kernel-source
    # hash matched the build that ran
    assert 0 < j["host_enqueue_ms_single_step"] < 0.25 * j["ms_per_step"] and r["traffic"] is not None and r["traffic"] >= r["algorithmic_mb_per_launch"]
