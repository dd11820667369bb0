import sys, json, time
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import bench_legs
from sonar_slam_amd import _lib
from sonar_slam_amd.CFAR import CFAR
ctx = _lib.default_context()
det = CFAR(40, 10, 0.1, 10)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
t = time.time()
o = bench_legs.chained(ctx, det, 16, n_sessions=S)
print(json.dumps(o, indent=1)); print("leg wall", time.time() - t)
