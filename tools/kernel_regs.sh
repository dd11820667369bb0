#!/bin/bash
# register / spill / scratch figures of every kernel in one HIP source (compile only, no GPU needed)
# usage: tools/kernel_regs.sh sonar_slam_amd/csrc/sfe_icp_sweep.hip
src="$1"
cd "$(dirname "$src")" || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 \
    -Rpass-analysis=kernel-resource-usage -c "$(basename "$src")" -o /dev/null 2>&1 |
    sed -n 's/.*remark: *//p' | sed 's/ \[-Rpass.*//' |
    awk '/^Function Name/ {if (line) print line; line=substr($3, 1, 48)} /^ *(VGPRs:|VGPRs Spill|SGPRs Spill|ScratchSize|Occupancy)/ {sub(/^ */, ""); line=line " | " $0} END {print line}'
