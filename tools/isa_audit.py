"""Static audit of the gfx950 ISA hipcc emits for the hand-scheduled kernels ("a guard that does not depend on hipcc's mood").

The hot kernels rely on properties that are invisible in the HIP source and that an innocent edit -- or a compiler upgrade -- can take away:

  * no scratch, no spilled registers (r03: 9 spilled registers cost bottleneck64_kernel<2> 7 %);
  * wave-uniform machinery (`v_readfirstlane_b32`, writes of M0 = the LDS-DMA destination, `buffer_load ... lds`, `s_barrier`) printed behind
    an EXEC mask is LISTED: `readfirstlane` takes the first ACTIVE lane and a masked LDS-DMA writes only the active lanes' bytes, so each
    such site is only correct if its condition is wave-uniform (today: all are `if (wave-derived value)` with an execz skip);
  * no VALU write of ANY store-of-more-than-64-bits' data registers within two wait states behind it, and every such store with an SGPR
    soffset followed by its `s_nop` guard (r04: the root cause of r03's intermittent wrong result -- hipcc exempts SGPR-soffset buffer stores from the store-data
    hazard and gfx950 does not; profiles/r04_store_data_hazard.txt), in EVERY kernel of the audited files, hot or not;
  * no divergent BRANCHES (`s_cbranch_execz / execnz`) between a kernel's first and last MFMA beyond the ones listed for it: r03's rule "no
    divergent control flow inside the phases of these kernels" (DESIGN 6c), checked instead of remembered.  Predication without a branch
    (`s_and_saveexec` ... `s_or_b64 exec`) is allowed and counted.

Usage: python tools/isa_audit.py            -> table for every kernel with MFMAs, exit 1 on a violation
       from tools.isa_audit import audit    -> tests/test_isa_audit_cpu.py
hipcc cross-compiles without a GPU (device-only, -S); conv.hip takes about a minute.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minddet_amd", "csrc")
OUT = os.path.join(CSRC, "build", "isa")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only"]

# kernel-name prefix -> exec branches allowed between the first and the last MFMA = the count in the reviewed ISA of r04.  Every one of them
# is a wave-UNIFORM decision on a thread-id-derived value that hipcc lowers through EXEC with an execz skip (`if (tid < 128) bias -> LDS`,
# `if (two)`: whole waves take it or skip it, EXEC inside is all ones): bottleneck64_kernel<0/1/2> 3 / 2 / 6, the fused-head ping-pong forms 1.
# A count above these means new divergent control flow inside a kernel's phases: read it before raising the number.
HOT = {
    "md::bottleneck64_kernel": 6,
    "md::conv_pingpong_kernel": 1,
    "md::conv1x1_stream_kernel": 0,
    "md::conv3x3_halo_kernel": 0,
    "md::stem_pool_kernel": 0,
    "md::stem_conv_kernel": 0,
    # c3pair: the `halo row inside the image` test of the T1 write-out (between the two MFMA phases: VALU + ds_write only), one per row fragment
    "md::c3pair32_kernel": 0,
    "md::c3pair64_kernel": 2,
    "md::c3pair128_kernel": 3,
}


def compile_isa(src):
    os.makedirs(OUT, exist_ok=True)
    dst = os.path.join(OUT, os.path.basename(src)[:-4] + ".s")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(ROOT, "include", "minddet_hip.h")]
    if not os.path.exists(dst) or os.path.getmtime(dst) < max(os.path.getmtime(d) for d in deps):
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", dst, src], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return dst


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def _meta(txt):
    """kernel name -> {vgpr_count, spill, scratch} from the YAML metadata block"""
    meta = {}
    for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm:
            continue
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", blk).group(1))
        meta[nm.group(1)] = dict(vgpr=g("vgpr_count"), spill=g("vgpr_spill_count"), sgpr_spill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"))
    return meta


_WIDE_STORE = re.compile(r"(buffer|global|flat|scratch)_store_dwordx[34] (?:\S+ )?v\[(\d+):(\d+)\]")
_SOFF = re.compile(r"buffer_store_dwordx[34] v\[\d+:\d+\], \S+ s\[\d+:\d+\], (s\d+|m0)\b")


def wide_store_hazards(lines):
    """-> list of findings for one kernel's instruction list: (a) a > 64-bit buffer store whose soffset is an SGPR (hipcc pads nothing
    behind it); (b) any > 64-bit store followed within two wait states by a VALU instruction that writes one of its data registers."""
    out = []
    for i, l in enumerate(lines):
        m = _WIDE_STORE.search(l)
        if not m or not l.startswith(m.group(1)):
            continue
        sgpr_soff = bool(_SOFF.match(l))   # the form hipcc pads nothing behind: the guard of MD_BUFFER_STORE_B128 must be there
        lo, hi = int(m.group(2)), int(m.group(3))
        ws = 0
        for l2 in lines[i + 1:i + 4]:
            if l2.startswith("s_nop"):
                ws += int(l2.split()[1]) + 1
            else:
                w = re.match(r"v_\w+ v(\d+)\b|v_\w+ v\[(\d+):(\d+)\]", l2)   # VALU writers only: an LDS / memory load returns tens of cycles later
                if w and ws < 2:
                    d0 = int(w.group(1) if w.group(1) is not None else w.group(2))
                    d1 = int(w.group(3)) if w.group(3) is not None else d0
                    if d0 <= hi and d1 >= lo:
                        out.append(f"data register rewritten {ws} wait state(s) behind the store: {l}  ->  {l2}")
                ws += 1
            if ws >= 2:
                break
        if sgpr_soff and not any(x.startswith("s_nop") for x in lines[i + 1:i + 12]):
            out.append("SGPR soffset without the s_nop guard behind it: " + l)
    return out


def audit_file(spath):
    txt = open(spath).read()
    meta = _meta(txt)
    res = {}
    for f in re.split(r"\n(?=_Z\w+:\s)", txt):
        m = re.match(r"(_Z\w+):", f)
        if not m:
            continue
        lines = [l.strip() for l in f.split("s_endpgm")[0].split("\n")]
        lines = [l for l in lines if l and not l.startswith(";") and not l.startswith(".")] + []
        code = [l for l in f.split("s_endpgm")[0].split("\n")]
        mf = [i for i, l in enumerate(lines) if l.startswith("v_mfma")]
        if not mf:
            sh = wide_store_hazards(lines)
            if sh:
                res[m.group(1)] = dict(store_hazards=sh, mfma=0, saveexec=0, exec_branches=0, inner_exec_branches=0, masked_uniform=[],
                                       **meta.get(m.group(1), dict(vgpr=-1, spill=-1, sgpr_spill=-1, scratch=-1)))
            continue
        # EXEC-mask depth along the printed order.  hipcc prints a structured region's blocks contiguously; a region entered by a branch
        # from elsewhere starts with its own saveexec, so a readfirstlane / M0 write / LDS-DMA at depth > 0 is inside SOME masked region.
        depth, masked_uniform = 0, []
        for i, l in enumerate(lines):
            if re.match(r"s_(and|andn2|or|xor)?_?saveexec_b64", l) and l.startswith("s_and_saveexec"):
                depth += 1
            elif l.startswith("s_or_b64 exec, exec"):
                depth = max(0, depth - 1)
            elif depth > 0 and (l.startswith("v_readfirstlane") or re.match(r"s_\w+ m0,", l) or (l.startswith("buffer_load") and l.endswith(" lds")) or
                                l.startswith("s_barrier")):
                masked_uniform.append(l)
        inner = lines[mf[0]:mf[-1] + 1]
        store_hazards = wide_store_hazards(lines)
        res[m.group(1)] = dict(store_hazards=store_hazards, mfma=len(mf), saveexec=sum("saveexec" in l for l in lines),
                               exec_branches=sum(l.startswith("s_cbranch_exec") for l in lines),
                               inner_exec_branches=sum(l.startswith("s_cbranch_exec") for l in inner),
                               masked_uniform=masked_uniform, **meta.get(m.group(1), dict(vgpr=-1, spill=-1, sgpr_spill=-1, scratch=-1)))
    return res


def audit(files=("bottleneck.hip", "c3pair.hip", "conv.hip", "stem.hip", "stemconv.hip", "detops.hip", "twostage.hip", "nms.hip", "pool.hip", "preproc.hip", "dcn.hip", "targets.hip")):
    """-> (rows, violations): rows = [(demangled name, stats)], violations = [str]"""
    rows, bad = [], []
    for f in files:
        r = audit_file(compile_isa(os.path.join(CSRC, f)))
        names = demangle(list(r))
        for k, st in r.items():
            d = names[k]
            rows.append((d, st))
            hot = [p for p in HOT if d.startswith("void " + p) or d.startswith(p)]
            # (the generic per-lane-K-walk instantiations of conv_igemm_kernel keep a 12-byte KWalk in private memory: not hot, reported only.
            # SGPR spills -- v_writelane / v_readlane -- exist in the persistent ping-pong forms, all outside the K loop: reported only.
            # A readfirstlane / M0 write / LDS-DMA printed behind a saveexec is reported only too: hipcc lowers wave-uniform `if (wave-derived)`
            # that way with an execz skip, which is correct -- it is listed so that a reader can check the condition IS wave-uniform.)
            for h in st["store_hazards"]:
                bad.append(f"{d}: store-data hazard: {h}")
            if hot and (st["spill"] or st["scratch"]):
                bad.append(f"{d}: {st['spill']} spilled vector registers, {st['scratch']} B of scratch")
            if hot and st["inner_exec_branches"] > HOT[hot[0]]:
                bad.append(f"{d}: {st['inner_exec_branches']} divergent branches between its first and last MFMA (allowed {HOT[hot[0]]})")
    return rows, bad


if __name__ == "__main__":
    rows, bad = audit()
    for d, st in rows:
        print(f"{d[:110]:110s} vgpr {st['vgpr']:3d} spill v{st['spill']} s{st['sgpr_spill']} scratch {st['scratch']:3d} mfma {st['mfma']:4d} saveexec {st['saveexec']:3d} "
              f"exec-branches {st['exec_branches']:3d} (inside MFMA span {st['inner_exec_branches']}) uniform-ops-behind-a-mask {len(st['masked_uniform'])}")
    for b in bad:
        print("VIOLATION", b)
    sys.exit(1 if bad else 0)
