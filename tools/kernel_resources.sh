#!/bin/bash
# Register / spill / scratch usage of every instantiation of a kernel in a csrc file (hipcc -Rpass-analysis=kernel-resource-usage).
# usage: bash tools/kernel_resources.sh conv.hip pingpong
set -e
cd "$(dirname "$0")/../minddet_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-function -fno-gpu-rdc -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kru.o 2>/tmp/kru.txt || { grep error /tmp/kru.txt | head; exit 1; }
python3 - "$2" <<'PY'
import re, sys
txt = open('/tmp/kru.txt').read()
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    if sys.argv[1] not in name: continue
    g = lambda k: re.search(k + r': (\d+)', b).group(1)
    print(name[:84], 'VGPR', g('VGPRs'), 'SGPRspill', g('SGPRs Spill'), 'VGPRspill', g('VGPRs Spill'), 'scratch', g(r'ScratchSize \[bytes/lane\]'))
PY
