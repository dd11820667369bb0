#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_matching_cost.py tests/test_global_init.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -6
timeout 200 python tools/default_flow_soak.py --seconds 45 --seed 3 2>&1 | tail -2
for v in many small; do
  if [ $v = small ]; then export SFE_COST_NO_MANY=1; fi
  timeout 1200 python tools/chained_leg.py 4096 4 > gpurun_out/r05_chained_replay_$v.json 2> gpurun_out/r05_chained_replay.err
  tail -2 gpurun_out/r05_chained_replay.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r05_chained_replay_$v.json"))
w = d["with_initialization"]
print("$v", {k: w[k] for k in ("keyframes_per_s", "seconds_per_run", "seconds_cost_table_launches", "scan_matches_replayed", "scan_matches_handed_to_scipy", "scan_matches_equal_to_the_scipy_only_run")}, w["parity"]["max_pose_diff_vs_oracle_chain"], "chained", d["keyframes_per_s"])
PY
done
