#!/bin/bash
# ISA of the bench's build of the ICP loop kernel alone (compile only, no GPU): register / spill figures and the static
# instruction mix; with a marker name, the region between `asm volatile("; NAME_BEGIN")` and `; NAME_END`.
# usage: tools/icp_isa.sh [out.s] [extra hipcc flags]
out=${1:-/tmp/icp_loop_bench.s}; shift
cd "$(dirname "$0")/../sonar_slam_amd/csrc" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -mllvm -amdgpu-atomic-optimizer-strategy=None \
    -DSW_INSPECT -Rpass-analysis=kernel-resource-usage -S --cuda-device-only -o "$out" "$@" sfe_icp_sweep_loop.hip 2>&1 |
    sed -n 's/.*remark: *//p' | sed 's/ \[-Rpass.*//' | grep -A12 "icp_sweep_kernel" | grep -E "VGPRs|SGPRs|Scratch|Occupancy" | tr '\n' '|'; echo
python3 - "$out" <<'PY'
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
st = [i for i, l in enumerate(lines) if l.startswith('_Z16icp_sweep_kernel')][0]
en = [i for i, l in enumerate(lines) if i > st and l.startswith('.Lfunc_end')][0]
c = collections.Counter()
for l in lines[st:en]:
    l = l.strip()
    if l and not l.startswith((';', '.')) and not l.endswith(':'):
        c[l.split()[0]] += 1
tot = sum(c.values())
spill = sum(v for k, v in c.items() if k.startswith('scratch_'))
print('instructions %d | v_readlane %d v_writelane %d s_nop %d scratch %d' % (tot, c['v_readlane_b32'], c['v_writelane_b32'], c['s_nop'], spill))
PY
