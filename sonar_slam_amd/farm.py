"""Multi-GPU job farm: independent scan-match / feature jobs, one worker per device.

The path shards with zero exchange: every ping's CFAR and every ICP job depends
only on its own inputs, so job j goes to device j mod G, each worker owns one ``sfe_ctx``
(one hipSetDevice, one stream, private scratch) and only the per-job results (3 x 3 transform,
status, iteration count) travel back over the host.  No RCCL collective, no xGMI traffic.

Two front ends:
  * ``IcpFarm``  -- one spawned process per device inside a single Python program
                    (the offline replay / batch tools use this).
  * ``shard`` / ``gather_results`` -- for programs already launched one rank per GPU by
                    ``torch.distributed.run`` (bench.py): ``torch.distributed`` is used for the
                    control plane only (barrier, gathering the tiny result records).
"""
import multiprocessing as mp
import os

import numpy as np

CHUNK = 1024  # scan matches per launch inside a farm worker


def shard(n_jobs, rank, world):
    """Static round-robin: the job indices rank ``rank`` of ``world`` processes."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, n_jobs, world))


def scatter_back(n_jobs, world, per_rank_results):
    """Inverse of ``shard``: per-rank result lists -> one list in job order."""
    out = [None] * n_jobs
    for rank, res in enumerate(per_rank_results):
        idx = shard(n_jobs, rank, world)
        if len(idx) != len(res):
            raise ValueError("rank %d returned %d results for %d jobs" % (rank, len(res), len(idx)))
        for j, r in zip(idx, res):
            out[j] = r
    return out


def gather_results(local_results, n_jobs):
    """All ranks call this with the results of ``shard(n_jobs, rank, world)`` (in that order);
    every rank gets the full list in job order.  Uses the default ``torch.distributed`` group
    (gloo or nccl); without an initialised group it is the identity for world size 1."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return scatter_back(n_jobs, 1, [list(local_results)])
    world = dist.get_world_size()
    buf = [None] * world
    dist.all_gather_object(buf, list(local_results))
    return scatter_back(n_jobs, world, buf)


def run_sharded(fn, n_jobs):
    """Evaluate ``fn(j)`` for this rank's shard and gather everything in job order."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(), dist.get_world_size()
    else:
        rank, world = 0, 1
    return gather_results([fn(j) for j in shard(n_jobs, rank, world)], n_jobs)


def _icp_worker(device, params_dict, jobs, conn):
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        from . import _lib, pcl
        ctx = _lib.Context(device)
        icp = pcl.ICP(ctx)
        icp.setParams(_lib.IcpParams(**params_dict))
        # flatten (pair, guess) into independent scan matches and run them CHUNK at a time in one launch
        flat = [(j, src, tgt, g) for j, (src, tgt, guesses) in enumerate(jobs) for g in guesses]
        res = []
        for c0 in range(0, len(flat), CHUNK):
            part = flat[c0:c0 + CHUNK]
            msgs, T, it = icp.compute_pairs([f[1] for f in part], [f[2] for f in part], [f[3] for f in part])
            res.extend(zip(msgs, T, it))
        out, k = [], 0
        for src, tgt, guesses in jobs:
            n = len(guesses)
            out.append(([r[0] for r in res[k:k + n]], np.stack([r[1] for r in res[k:k + n]]) if n else
                        np.zeros((0, 3, 3), np.float32), np.array([r[2] for r in res[k:k + n]], np.int32)))
            k += n
        conn.send(("ok", out))
    except Exception as e:  # surfaced in the parent, never swallowed
        conn.send(("error", "%s: %s" % (type(e).__name__, e)))
    finally:
        conn.close()


class IcpFarm(object):
    """Farm (source, target, [guesses]) jobs over ``devices`` (default: every visible GPU)."""

    def __init__(self, params, devices=None):
        from . import _lib
        self.params = params
        if devices is None:
            devices = list(range(_lib.device_count()))
        if not devices:
            raise _lib.SonarFEError("IcpFarm: no HIP device visible; there is no CPU fallback")
        self.devices = list(devices)

    def run(self, jobs):
        """jobs: list of (source Nx2, target Mx2, guesses [k x 3 x 3]).  Returns, in job order,
        (messages [k], T [k x 3 x 3], iterations [k])."""
        world = len(self.devices)
        ctxm = mp.get_context("spawn")
        procs = []
        for rank, dev in enumerate(self.devices):
            mine = [jobs[j] for j in shard(len(jobs), rank, world)]
            parent, child = ctxm.Pipe()
            p = ctxm.Process(target=_icp_worker, args=(dev, self.params.as_dict(), mine, child))
            p.start()
            procs.append((p, parent))
        per_rank = []
        for p, parent in procs:
            tag, payload = parent.recv()
            p.join()
            if tag != "ok":
                raise RuntimeError("IcpFarm worker failed: %s" % payload)
            per_rank.append(payload)
        return scatter_back(len(jobs), world, per_rank)
