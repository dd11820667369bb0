#!/usr/bin/env python
"""The bench line's `chained` leg on its own (tools/bench_legs.py): S sessions x 8 keyframes on device-resident clouds.
usage: chained_leg.py [sessions] [parity_sessions]   -- e.g. `chained_leg.py 64 64` checks EVERY session against the
oracle's chain (oracle/chain.py)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from sonar_slam_amd import _lib  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402

ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
P = int(sys.argv[2]) if len(sys.argv) > 2 else 4
t = time.time()
o = bench_legs.chained(ctx, det, os.cpu_count() or 1, n_sessions=S, n_distinct=min(S, 64), parity_sessions=min(P, S, 64))
o["leg_wall_s"] = time.time() - t
print(json.dumps(o, indent=1))
