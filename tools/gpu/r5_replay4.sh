#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_global_init.py -m gpu -q -x --tb=short -p no:cacheprovider 2>&1 | tail -6
timeout 1200 python tools/chained_leg.py 4096 4 > gpurun_out/r05_chained_replay.json 2> gpurun_out/r05_chained_replay.err
tail -3 gpurun_out/r05_chained_replay.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_chained_replay.json"))
w = d["with_initialization"]
print({k: w[k] for k in w if k not in ("note", "parity", "scipy_only")})
print(w.get("parity"))
print("chained", d["keyframes_per_s"], "leg wall", d["leg_wall_s"])
PY
