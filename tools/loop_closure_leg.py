#!/usr/bin/env python
"""The bench line's `loop_closure` leg on its own (tools/bench_legs.py)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs  # noqa: E402
from sonar_slam_amd import _lib  # noqa: E402
from sonar_slam_amd.CFAR import CFAR  # noqa: E402

ctx = _lib.default_context()
print(json.dumps(bench_legs.loop_closure(ctx, CFAR(40, 10, 0.1, 10), os.cpu_count() or 1), indent=1))
