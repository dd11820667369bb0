#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_global_init.py tests/test_matching_cost.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -6
timeout 900 python tools/loop_closure_leg.py > gpurun_out/r05_loop_closure_replay.json 2> gpurun_out/r05_loop_closure_replay.err
tail -3 gpurun_out/r05_loop_closure_replay.err
python -c "
import json; d=json.load(open('gpurun_out/r05_loop_closure_replay.json')); print({k:d[k] for k in d if k not in ('note','parity','workload')})"
