"""Multi-GPU job farm: independent scan-match / feature jobs, one worker per device.

The path shards with zero exchange: every ping's CFAR and every ICP job depends
only on its own inputs, so job j goes to device j mod G, each worker owns one ``sfe_ctx``
(one hipSetDevice, one stream, private scratch) and only the per-job results (3 x 3 transform,
status, iteration count) travel back over the host.  No RCCL collective, no xGMI traffic.

Two front ends:
  * ``IcpFarm``  -- one PERSISTENT worker process per device inside a single Python program
                    (the offline replay / batch tools use this): the workers are spawned once, keep
                    their ``sfe_ctx`` (and its scratch) alive between calls, and receive every batch
                    of jobs through a shared-memory block -- only a small layout record travels
                    over the pipe, no cloud is pickled.
  * ``shard`` / ``gather_results`` -- for programs already launched one rank per GPU by
                    ``torch.distributed.run`` (bench.py): ``torch.distributed`` is used for the
                    control plane only (barrier, gathering the tiny result records).
"""
import importlib
import multiprocessing as mp
import os
from multiprocessing import shared_memory

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


# ---- shared-memory job blocks ---------------------------------------------------------------
# One block per worker, written by the parent, read (clouds, job table, guesses) and written (results) by
# the worker:   [src pool f32 Ns x 2][tgt pool f32 Nt x 2][jobs4 i32 n x 4][guesses f32 n x 9]
#               [T f32 n x 9][status i32 n][iters i32 n]           every array 64-byte aligned.
# A cloud object that several jobs name (many guesses on one pair, one target matched against many
# sources, a cycled set of distinct pairs) is stored once; the job table refers to it by offset.
_ALIGN = 64


def _layout(ns_pts, nt_pts, n):
    off, out = 0, {}
    for name, nbytes in (("src", 8 * ns_pts), ("tgt", 8 * nt_pts), ("jobs4", 16 * n), ("guess", 36 * n),
                         ("T", 36 * n), ("status", 4 * n), ("iters", 4 * n)):
        out[name] = off
        off = (off + nbytes + _ALIGN - 1) // _ALIGN * _ALIGN
    out.update(ns_pts=ns_pts, nt_pts=nt_pts, n=n, bytes=max(off, _ALIGN))
    return out


def _views(buf, lay):
    n = lay["n"]

    def arr(name, dtype, shape):
        return np.ndarray(shape, dtype, buffer=buf, offset=lay[name])
    return {"src": arr("src", np.float32, (lay["ns_pts"], 2)), "tgt": arr("tgt", np.float32, (lay["nt_pts"], 2)),
            "jobs4": arr("jobs4", np.int32, (n, 4)), "guess": arr("guess", np.float32, (n, 9)),
            "T": arr("T", np.float32, (n, 3, 3)), "status": arr("status", np.int32, (n,)),
            "iters": arr("iters", np.int32, (n,))}


def _cloud32(a, what):
    a = np.asarray(a)
    if a.ndim != 2 or a.shape[1] != 2:
        raise TypeError("IcpFarm: %s must be an N x 2 array, got shape %r" % (what, a.shape))
    if len(a) == 0:
        raise RuntimeError("IcpFarm: empty %s cloud (libpointmatcher would throw)" % what)
    return a


def pack_jobs(jobs):
    """jobs: list of (source, target, guesses).  -> (pools, table): the distinct clouds in first-use order with
    their offsets, and the flat (job, guess) table.  Pure host logic (tested without a device)."""
    src_pool, tgt_pool, src_at, tgt_at = [], [], {}, {}
    ns_pts = nt_pts = 0
    rows, guesses = [], []
    for src, tgt, gs in jobs:
        src, tgt = _cloud32(src, "source"), _cloud32(tgt, "target")
        if id(src) not in src_at:
            src_at[id(src)] = ns_pts
            src_pool.append(src)
            ns_pts += len(src)
        if id(tgt) not in tgt_at:
            tgt_at[id(tgt)] = nt_pts
            tgt_pool.append(tgt)
            nt_pts += len(tgt)
        for g in gs:
            g = np.asarray(g, np.float32)
            if g.shape != (3, 3):
                raise TypeError("IcpFarm: guess must be 3 x 3, got %r" % (g.shape,))
            rows.append((src_at[id(src)], len(src), tgt_at[id(tgt)], len(tgt)))
            guesses.append(g.reshape(9))
    return src_pool, tgt_pool, ns_pts, nt_pts, rows, guesses


def _chunk_slab(jobs4, lo_col, n_col):
    """the contiguous slab of a pool that a chunk of the job table refers to"""
    lo = int(jobs4[:, lo_col].min())
    hi = int((jobs4[:, lo_col] + jobs4[:, n_col]).max())
    return lo, hi


def _hip_compute(device, params_dict):
    """the product backend of a farm worker: one sfe_ctx on `device`, chunks of the job table through
    sfe_icp_compute_jobs (no CPU fallback: without the library or a gfx950 device this raises)"""
    from . import _lib, pcl
    ctx = _lib.Context(device)
    icp = pcl.ICP(ctx)
    icp.setParams(_lib.IcpParams(**params_dict))

    def run(v, chunk):
        n = len(v["jobs4"])
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            j4 = v["jobs4"][c0:c1].copy()
            # only the slabs of the pools this chunk names travel to the device
            s_lo, s_hi = _chunk_slab(j4, 0, 1)
            t_lo, t_hi = _chunk_slab(j4, 2, 3)
            j4[:, 0] -= s_lo
            j4[:, 2] -= t_lo
            icp.compute_jobs(v["src"][s_lo:s_hi], v["tgt"][t_lo:t_hi], j4, v["guess"][c0:c1],
                             out=(v["status"][c0:c1], v["T"][c0:c1], v["iters"][c0:c1]))
    return ctx.name(), run


def _farm_worker(device, params_dict, backend, conn):
    """Persistent worker: create the backend once, then serve ("run", shm name, layout, chunk) requests until
    ("stop",).  Every failure is reported to the parent, never swallowed."""
    shm = None
    try:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            name, run = _hip_compute(device, params_dict)
        else:  # test hook: "module:function" -> (name, run) like _hip_compute
            mod, fn = backend.split(":")
            name, run = getattr(importlib.import_module(mod), fn)(device, params_dict)
        conn.send(("ready", name))
        while True:
            msg = conn.recv()
            if msg[0] == "stop":
                break
            _, shm_name, lay, chunk, seq = msg
            try:
                if shm is None or shm.name != shm_name:
                    if shm is not None:
                        shm.close()
                    shm = shared_memory.SharedMemory(name=shm_name)
                run(_views(shm.buf, lay), chunk)
                conn.send(("ok", lay["n"], seq))
            except Exception as e:
                conn.send(("error", "%s: %s" % (type(e).__name__, e), seq))
    except Exception as e:
        try:
            conn.send(("error", "%s: %s" % (type(e).__name__, e)))
        except Exception:
            pass
    finally:
        if shm is not None:
            try:
                shm.close()
            except Exception:
                pass
        conn.close()


class _Worker(object):
    def __init__(self, proc, conn, device):
        self.proc, self.conn, self.device = proc, conn, device
        self.shm = None
        self.name = None
        self.pending = None     # sequence number of a request this worker has not answered yet

    def block(self, nbytes):
        """this worker's shared-memory block, grown (never shrunk) to hold nbytes"""
        if self.shm is None or self.shm.size < nbytes:
            if self.shm is not None:
                self.shm.close()
                self.shm.unlink()
            self.shm = shared_memory.SharedMemory(create=True, size=int(nbytes + nbytes // 4 + 4096))
        return self.shm


class IcpFarm(object):
    """Farm (source, target, [guesses]) jobs over ``devices`` (default: every visible GPU): job j -> worker
    j mod G.  The worker processes start on the first ``run`` (or ``start()``) and live until ``close()``;
    use it as a context manager."""

    def __init__(self, params, devices=None, chunk=CHUNK, _backend=None):
        from . import _lib
        self.params = params
        if devices is None:
            devices = list(range(_lib.device_count()))
        if not devices:
            raise _lib.SonarFEError("IcpFarm: no HIP device visible; there is no CPU fallback")
        self.devices = list(devices)
        self.chunk = int(chunk)
        self._backend = _backend
        self._workers = []
        self._seq = 0           # request number, echoed by the worker: a stale reply can never answer a newer request

    def start(self):
        if self._workers:
            return self
        ctxm = mp.get_context("spawn")      # a HIP context does not survive fork()
        for dev in self.devices:
            parent, child = ctxm.Pipe()
            p = ctxm.Process(target=_farm_worker, args=(dev, self.params.as_dict(), self._backend, child), daemon=True)
            p.start()
            child.close()
            self._workers.append(_Worker(p, parent, dev))
        try:
            for w in self._workers:
                tag, payload = self._recv(w)
                if tag != "ready":
                    raise RuntimeError("IcpFarm worker on device %d failed to start: %s" % (w.device, payload))
                w.name = payload
        except Exception:
            self.close()
            raise
        return self

    @staticmethod
    def _recv(w):
        try:
            return w.conn.recv()
        except EOFError:
            return "error", "worker process on device %d died (exit code %r)" % (w.device, w.proc.exitcode)

    def run(self, jobs):
        """jobs: list of (source Nx2, target Mx2, guesses [k x 3 x 3]).  Returns, in job order,
        (messages [k], T [k x 3 x 3], iterations [k])."""
        from . import _lib
        self.start()
        world = len(self._workers)
        # 1. validate and pack EVERY rank's jobs before any worker hears of this batch: a bad job (empty cloud, guess of
        #    the wrong shape) raises here, with nothing in flight (a request left unanswered used to be read
        #    as the reply of the NEXT run()).
        packed = []
        for rank in range(world):
            mine = [jobs[j] for j in shard(len(jobs), rank, world)]
            packed.append((mine, pack_jobs(mine)))
        self._seq += 1
        seq = self._seq

        def fill_and_send(w, mine, pk):
            """one rank's block: fill the shared memory, start the worker; -> what the collection below needs"""
            src_pool, tgt_pool, ns_pts, nt_pts, rows, guesses = pk
            if w.pending is not None:
                # a request abandoned by an interrupted run (KeyboardInterrupt between send and reply) may still be
                # executing: its worker reads -- and writes results into -- the block about to be refilled.
                # Its reply is awaited (and dropped) before the block is touched.
                self._recv_reply(w, w.pending)
            lay = _layout(ns_pts, nt_pts, len(rows))
            v = _views(w.block(lay["bytes"]).buf, lay)
            o = 0
            for c in src_pool:                      # one copy (and the float32 cast) per DISTINCT cloud
                v["src"][o:o + len(c)] = c
                o += len(c)
            o = 0
            for c in tgt_pool:
                v["tgt"][o:o + len(c)] = c
                o += len(c)
            if rows:
                v["jobs4"][:] = np.asarray(rows, np.int32)
                v["guess"][:] = np.stack(guesses)
            del v
            w.conn.send(("run", w.shm.name, lay, self.chunk, seq))
            w.pending = seq
            return w, lay, [len(gs) for _, _, gs in mine]

        # 2. fill the shared-memory blocks and start the workers: one packing thread per rank (the copies into shared
        #    memory are plain numpy slice assignments, which release the GIL) -- with G devices the parent would
        #    otherwise pack G blocks one after the other while G - 1 workers wait for theirs.  A rank's worker starts
        #    as soon as its own block is complete.
        sent, errors = [], []
        if world == 1:
            try:
                sent.append(fill_and_send(self._workers[0], *packed[0]))
            except BaseException as e:   # noqa: B902 (re-raised below, after the replies are drained)
                errors.append(e)
        else:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(world, 16)) as pool:
                futs = [pool.submit(fill_and_send, w, mine, pk) for w, (mine, pk) in zip(self._workers, packed)]
                for f in futs:
                    try:
                        sent.append(f.result())
                    except BaseException as e:   # noqa: B902
                        errors.append(e)
        if errors:
            # something failed after some requests went out: every request already out is answered before the error
            # leaves, so no worker is still reading a block (or owes a reply) when the caller tries again
            for w, _, _ in sent:
                self._recv_reply(w, seq)
            raise errors[0]
        per_rank, failure = [], None
        for w, lay, ks in sent:
            tag, payload = self._recv_reply(w, seq)
            if tag != "ok":
                failure = failure or "device %d: %s" % (w.device, payload)
                continue
            v = _views(w.shm.buf, lay)
            out, k0 = [], 0
            for k in ks:
                st = v["status"][k0:k0 + k]
                out.append(([_lib.ICP_STATUS_MESSAGES.get(int(s), "ICP failure %d" % s) for s in st],
                            v["T"][k0:k0 + k].copy(), v["iters"][k0:k0 + k].copy()))
                k0 += k
            del v
            per_rank.append(out)
        if failure:
            raise RuntimeError("IcpFarm worker failed: %s" % failure)
        return scatter_back(len(jobs), world, per_rank)

    def _recv_reply(self, w, seq):
        """the reply to request `seq` of worker w; a reply carrying an older number (a request abandoned by an
        interrupted run) is discarded, never taken for this one"""
        while True:
            msg = self._recv(w)
            if msg[0] == "error" and len(msg) == 2:       # worker died / start-up failure: no sequence number
                w.pending = None
                return msg
            if msg[2] == seq:
                w.pending = None
                return msg[0], msg[1]

    def close(self):
        for w in self._workers:
            try:
                w.conn.send(("stop",))
            except Exception:
                pass
        for w in self._workers:
            w.proc.join(timeout=10)
            if w.proc.is_alive():
                w.proc.terminate()      # this exact process, by handle
                w.proc.join(timeout=5)
            try:
                w.conn.close()
            except Exception:
                pass
            if w.shm is not None:
                try:
                    w.shm.close()
                    w.shm.unlink()
                except Exception:
                    pass
                w.shm = None
        self._workers = []

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
