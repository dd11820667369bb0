#!/usr/bin/env python
"""BASELINE configs[2] stand-in: offline replay of a synthetic ping sequence (no rosbag / gtsam in
this image) through the GPU front end; prints per-keyframe scan-match results, the pose error of
the ICP chain vs the drifting odometry, and front-end throughput.  `--save x.npz` writes the ping
sequence in the harness's format (images, bearings, range_resolution, stamps, odom, truth);
`--load x.npz` replays one (e.g. converted from a real bag on a machine that has rosbag)."""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_slam_amd import _lib, synth  # noqa: E402
from sonar_slam_amd.feature_extraction import FeatureExtraction, SonarPing, oculus_bearings  # noqa: E402
from sonar_slam_amd.pose2 import Pose2  # noqa: E402
from sonar_slam_amd.replay import FrontEnd, replay  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pings", type=int, default=40)
    ap.add_argument("--rows", type=int, default=1024)
    ap.add_argument("--beams", type=int, default=512)
    ap.add_argument("--store", action="store_true", help="keyframe clouds stay on the device (CloudStore): no wire hop")
    ap.add_argument("--no-initialization", action="store_true",
                    help="scan matches start from the odometry (slam.py:665-666 with the flag off) instead of the shgo global "
                         "initialisation the reference runs by default (slam.py:77)")
    ap.add_argument("--loop-closures", action="store_true", help="the loop-closure search of slam.py:839-1087 after every keyframe")
    ap.add_argument("--turn", type=float, default=0.03, help="heading change per ping of the synthetic trajectory (2 pi / N closes a loop)")
    ap.add_argument("--save")
    ap.add_argument("--load")
    a = ap.parse_args()
    if a.load:
        z = np.load(a.load)
        images, bearings, res, stamps, dr, true = (z["images"], z["bearings"], float(z["range_resolution"]),
                                                   z["stamps"], z["odom"], z["truth"] if "truth" in z else None)
    else:
        world = synth.world_structure(seed=2, n=12000)
        true, dr = synth.trajectory(n=a.pings, step=0.9, turn=a.turn, seed=3, start=(20.0, 0.0, 0.0) if a.turn > 0.1 else (2.0, 0.0, 0.0))
        bearings = oculus_bearings(a.beams)
        images = np.stack([synth.render_ping(world, p, bearings, rows=a.rows, seed=i) for i, p in enumerate(true)])
        res, stamps = 30.0 / a.rows, np.arange(a.pings, dtype=float)
    if a.save:
        np.savez_compressed(a.save, images=images, bearings=bearings, range_resolution=res, stamps=stamps, odom=dr,
                            truth=true)
    ctx = _lib.default_context()
    pings = [SonarPing(im, bearings, res, ping_id=i) for i, im in enumerate(images)]
    fe = FeatureExtraction(ctx)
    fe.Ntc, fe.Ngc, fe.Pfa, fe.rank, fe.alg, fe.threshold = 40, 10, 0.1, 10, "SOCA", 65
    fe.resolution, fe.outlier_filter_radius, fe.outlier_filter_min_points, fe.skip = 0.5, 1.0, 5, 1
    fe.configure()
    fe.callback(pings[0])      # warm-up: maps + device geometry
    store = None
    if a.store:
        from sonar_slam_amd.store import CloudStore
        store = CloudStore(ctx, capacity_points=1 << 20, max_clouds=4096)
    front = FrontEnd(ctx, keyframe_translation=1.5, keyframe_duration=0.5, store=store, ssm_initialization=not a.no_initialization,
                     nssm_enable=a.loop_closures)
    log, t_fe, t_slam = replay(pings, stamps, dr, fe, front)
    print("%d pings, %d keyframes; feature extraction %.2f ms/ping, SLAM front end %.2f ms/keyframe (host wall, "
          "single-ping host API incl. PCIe copies%s)" % (len(pings), len(log), 1e3 * t_fe / len(pings),
                                                          1e3 * t_slam / max(len(log), 1),
                                                          "; keyframe clouds device-resident" if a.store else ""))
    for r in log:
        k = int(r["time"])
        line = "kf %2d ping %3d %-20s src %5d tgt %5d" % (r["source_key"], k, r["status"], r.get("n_source", 0),
                                                          r.get("n_target", 0))
        if true is not None:
            want = Pose2(*true[0]).between(Pose2(*true[k]))
            got = Pose2(*dr[0]).between(Pose2(*r["pose"]))
            odo = Pose2(*dr[0]).between(Pose2(*dr[k]))
            line += "  |err| icp-chain %.3f m  odometry %.3f m" % (np.hypot(got.x() - want.x(), got.y() - want.y()),
                                                                   np.hypot(odo.x() - want.x(), odo.y() - want.y()))
        if "init_x" in r:
            line += "  shgo x (%+.3f %+.3f %+.4f) cost %d" % (r["init_x"] + (int(r["init_cost"]),))
        n = r.get("nssm")
        if n is not None:
            line += "  | loop search: %s" % n["status"]
            if "target_key" in n:
                line += " -> kf %d, %d/%d ICPs converged" % (n["target_key"], n.get("n_converged", 0), n.get("n_guesses", 0))
        print(line)


if __name__ == "__main__":
    main()
