"""Helper of test_host.test_gloo_farm: one rank of a W-process gloo job farm (W = 2 and the target machine's 8)."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (tests may use the oracle as the compute stand-in)
from sonar_slam_amd import farm, synth  # noqa: E402


def job(j):
    src, tgt, guess, _ = synth.scan_pair(seed=j, n_src=200, n_tgt=200)
    st, T, it = oracle.icp(src, tgt, guess)
    return (st, T.tobytes(), it)


def main():
    dist.init_process_group("gloo")
    w = dist.get_world_size()
    n_jobs = 7 if w == 2 else 2 * w + 3          # (not a multiple of the world size: the shards differ in length)
    res = farm.run_sharded(job, n_jobs)
    serial = [job(j) for j in range(n_jobs)]
    assert res == serial, "sharded results differ from serial ones"
    mine = farm.shard(n_jobs, dist.get_rank(), w)
    assert mine == list(range(dist.get_rank(), n_jobs, w)) and len(mine) in (n_jobs // w, n_jobs // w + 1)  # job j -> rank j mod W
    dist.barrier()
    dist.destroy_process_group()
    print("FARM_OK rank", os.environ["RANK"])


if __name__ == "__main__":
    main()
