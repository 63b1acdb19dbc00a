#!/bin/bash
# Diagnostic builds for the r03 "wrong only sometimes" result of bottleneck64_kernel (DESIGN 6c lesson 5).
# Generates variants of csrc/bottleneck.hip that differ ONLY in how the lane -> pixel map is written (hp_now) and in the
# ordering point between the slab writes and the read-out, compiles each to ISA (.s kept for reading) and links it with the
# product's other objects into minddet_amd/csrc/build/variants/lib_<name>.so.  tools/bn_chain_repro.py runs them.
# Runs in the build container (hipcc cross-compiles); the .so files travel to the GPU box with the tree (build/ is git-ignored).
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
SRC=$HERE/minddet_amd/csrc
OUT=$SRC/build/variants
mkdir -p "$OUT"
make -s -C "$SRC" -j8
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fno-gpu-rdc"
python3 - "$SRC/bottleneck.hip" "$OUT" <<'EOF'
import sys
src, out = sys.argv[1], sys.argv[2]
s = open(src).read().replace('#include "aot.h"', '#include "../../aot.h"')
MAP = "return (int)(((0x73261540u >> ((l >> 2) * 4)) & 7u) << 2) | (l & 3);"
assert s.count(MAP) == 1
variants = {
    "cur": s,
    # the r03 first version as DESIGN 6c lesson 5 describes it: a select chain behind the opaque copy
    "chain": s.replace(MAP, "const int b = l >> 2;\n        const int nb = b == 1 ? 4 : b == 2 ? 5 : b == 3 ? 1 : b == 4 ? 6 : b == 5 ? 2 : b == 6 ? 3 : b;\n        return (nb << 2) | (l & 3);"),
    "switch": s.replace(MAP, "int nb; switch (l >> 2) { case 1: nb = 4; break; case 2: nb = 5; break; case 3: nb = 1; break; case 4: nb = 6; break; case 5: nb = 2; break; case 6: nb = 3; break; default: nb = l >> 2; }\n        return (nb << 2) | (l & 3);"),
}
for name, text in variants.items():
    open(f"{out}/bottleneck_{name}.hip", "w").write(text)
EOF
for v in cur chain switch; do
    /opt/rocm/bin/hipcc $FLAGS -S --cuda-device-only -o "$OUT/bottleneck_$v.s" "$OUT/bottleneck_$v.hip" 2>/dev/null
    /opt/rocm/bin/hipcc $FLAGS -c "$OUT/bottleneck_$v.hip" -o "$OUT/bottleneck_$v.o"
    OBJS=$(ls "$SRC"/build/*.o | grep -v '/diag_' | grep -v '/bottleneck.o')
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT/lib_$v.so" $OBJS "$OUT/bottleneck_$v.o"
    echo "$v: $(grep -c 's_and_saveexec\|s_cbranch_exec' "$OUT/bottleneck_$v.s") exec-mask instructions"
done
