#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export SFE_SG_PIECE=4
cd /tmp && export TMPDIR=/tmp
timeout 60 rocprofv3 -L 2>/dev/null | grep -o "\b\(TCP\|TA\|TD\|TCC\|SQ\)_[A-Za-z0-9_]*" | sort -u > $GRAFT_REPO_ROOT/gpurun_out/counters.txt
wc -l $GRAFT_REPO_ROOT/gpurun_out/counters.txt
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $GRAFT_REPO_ROOT/gpurun_out/pmc_a -- python $GRAFT_REPO_ROOT/tools/extract_times.py 512 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT -d $GRAFT_REPO_ROOT/gpurun_out/pmc_b -- python $GRAFT_REPO_ROOT/tools/extract_times.py 512 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TCP_TA_TCP_STATE_READ_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum -d $GRAFT_REPO_ROOT/gpurun_out/pmc_c -- python $GRAFT_REPO_ROOT/tools/extract_times.py 512 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c 2>&1 | head
