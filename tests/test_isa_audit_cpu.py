"""CPU (hipcc cross-compiles without a GPU): the static ISA audit of the hand-scheduled kernels, tools/isa_audit.py -- no spilled vector
registers / scratch in the hot kernels, and no divergent branch between a hot kernel's first and last MFMA beyond the reviewed ones
(DESIGN 6c lesson 5 / 6d: "no divergent control flow inside the phases of these kernels", checked instead of remembered)."""
import os
import shutil
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc + c++filt")
def test_hot_kernels_pass_the_isa_audit():
    from tools.isa_audit import HOT, audit

    rows, bad = audit()
    assert not bad, "\n".join(bad)
    seen = {p: 0 for p in HOT}
    for d, st in rows:
        for p in HOT:
            if p in d:
                seen[p] += 1
                assert st["vgpr"] <= 256 and st["mfma"] > 0
    assert all(n > 0 for n in seen.values()), seen     # every hot kernel family was found in the ISA (a renamed kernel would escape the audit)
    # the two register budgets the occupancy of these kernels rests on: bottleneck64_kernel at 4 waves / SIMD, the ping-pong kernel at 2
    for d, st in rows:
        if "md::bottleneck64_kernel" in d:
            assert st["vgpr"] <= 128, (d, st["vgpr"])
