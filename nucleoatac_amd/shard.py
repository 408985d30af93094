"""Chunk-list sharding across the GPUs of one node (one process per GPU, no data-path collective).

The reference parallelises with multiprocessing.Pool.map over chunks and consumes results in chunk order
(nucleoatac/run_occ.py:101-123, run_nuc.py:164-188).  Chunks are independent, so the MI355X version splits the
(sorted, merged) chunk list into `world` contiguous ranges balanced by bases + kappa * fragments; every rank
runs the whole pipeline on its range with its own natac context, and only small per-chunk results travel
(torch.distributed gather over RCCL/gloo) -- per-base tracks are written by the rank that produced them, in
rank order == chunk order.
"""
import os

import numpy as np


def balanced_ranges(chunk_len, frag_off, world, kappa=4.0):
    """`world` contiguous chunk ranges [(lo, hi), ...] with ~equal cost sum(L) + kappa * sum(F)."""
    chunk_len = np.asarray(chunk_len, dtype=np.int64)
    nfr = np.diff(np.asarray(frag_off, dtype=np.int64))
    cost = chunk_len + kappa * nfr
    cum = np.concatenate(([0], np.cumsum(cost, dtype=np.float64)))
    total = cum[-1]
    nc = len(chunk_len)
    cuts = [0]
    for r in range(1, world):
        j = int(np.searchsorted(cum, total * r / world, side="left"))
        j = min(max(j, cuts[-1]), nc)
        cuts.append(j)
    cuts.append(nc)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


_STATE = {"gpu_group": None, "backend": None}
GROUP_TIMEOUT_S = 12 * 3600      # control plane (gloo): single-rank phases (merge, a slow shard) may hold the others for a long time
RCCL_TIMEOUT_S = 600             # RCCL group: only the benchmark's timing barrier runs over it (ranks arrive together)


def ensure_distributed(prefer=None, device=None):
    """Under torchrun (WORLD_SIZE > 1) make sure the process group exists before any rank-dependent work: run_occ /
    run_nuc called through the API (not cli.main) would otherwise skip the barrier and the gather silently and rank 0
    would merge part files that other ranks are still writing.

    The control plane -- barriers, gathers of small per-chunk objects, the ok / failed flag of single-rank steps -- is a gloo
    group (CPU, always available) with a long timeout; there is no data-path collective, so RCCL is OPT-IN: when `prefer`
    (default: the NATAC_DIST_BACKEND environment variable, else "gloo") is "nccl", an RCCL subgroup is created on top of it and probed; whether
    it is used is decided COLLECTIVELY (a MIN all-reduce of the per-rank probe results over gloo), so one rank with a broken
    RCCL cannot leave the others waiting in an RCCL collective for long (its own timeout is 10 minutes).  The drivers' barriers
    always use gloo; `barrier(sync_cuda=True)` -- the benchmark's timing barrier -- synchronises over RCCL when all ranks have
    it.  Returns (dist module or None, True if this call created the group)."""
    rank, world, local = env_rank_world()
    if world <= 1:
        return None, False
    import torch.distributed as dist
    if dist.is_initialized():
        if dist.get_world_size() != world:
            raise RuntimeError("WORLD_SIZE=%d but the torch.distributed group has %d ranks" % (world, dist.get_world_size()))
        return dist, False
    import datetime
    import sys
    import torch
    prefer = prefer or os.environ.get("NATAC_DIST_BACKEND", "gloo")
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=GROUP_TIMEOUT_S))
    _STATE["backend"] = "gloo"
    if prefer == "nccl":
        dev = int(device if device is not None else os.environ.get("NATAC_DEVICE", local))   # the GPU this rank computes on
        ok, why, grp = 1, "", None
        try:
            if not torch.cuda.is_available():
                raise RuntimeError("no GPU visible to torch")
            torch.cuda.set_device(dev)
        except Exception as e:      # noqa: BLE001
            ok, why = 0, "%s: %s" % (type(e).__name__, str(e)[:200])
        flag = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)                   # does every rank have a usable GPU + torch runtime?
        if int(flag.item()) == 1:
            # ranks that share one GPU (functional tests on a 1-GPU box) cannot form an RCCL communicator: agree on that too
            devs = [None] * world
            dist.all_gather_object(devs, dev)
            if len(set(devs)) < world:
                ok, why = 0, "ranks share GPUs %s" % devs
            else:
                try:
                    grp = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=RCCL_TIMEOUT_S))
                    probe = torch.ones(1, device="cuda")
                    dist.all_reduce(probe, group=grp)
                    torch.cuda.synchronize()
                    if int(probe.item()) != world:
                        raise RuntimeError("all_reduce probe returned %r" % probe.item())
                except Exception as e:      # noqa: BLE001
                    ok, why = 0, "%s: %s" % (type(e).__name__, str(e)[:200])
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            _STATE["gpu_group"], _STATE["backend"] = grp, "nccl"
        elif rank == 0 or why:
            sys.stderr.write("nucleoatac_amd: RCCL group not used (%s); barriers over gloo\n" % (why or "another rank could not join"))
    return dist, True


def control_backend():
    """"nccl" when barrier() synchronises over RCCL, "gloo" otherwise, None without a group"""
    return _STATE["backend"]


def barrier(sync_cuda=False):
    """all ranks reach this point (no-op on one rank; raises if WORLD_SIZE > 1 without a process group).  sync_cuda: the
    benchmark's form -- torch.cuda.synchronize() + a barrier over the RCCL group when the ranks agreed on one; the drivers'
    plain barriers run over gloo, whose timeout allows for ranks that arrive minutes apart."""
    dist, _ = ensure_distributed()
    if dist is not None:
        if sync_cuda and _STATE["gpu_group"] is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier(group=_STATE["gpu_group"])
        else:
            dist.barrier()


def all_reduce_scalars(values, op="sum"):
    """element-wise sum / max of a short list of floats over the ranks (control plane; identity on one rank)"""
    dist, _ = ensure_distributed()
    if dist is None:
        return [float(v) for v in values]
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == "max" else dist.ReduceOp.SUM)
    return [float(x) for x in t.tolist()]


def run_on_rank0(fn, *args):
    """single-rank step (vprocess, merge): rank 0 runs `fn`, the other ranks wait for its ok / failed flag -- a failure on
    rank 0 raises on every rank instead of leaving them in a barrier until the group times out."""
    rank, world, _ = env_rank_world()
    if world <= 1:
        return fn(*args)
    dist, _ = ensure_distributed()
    import torch
    err = None
    out = None
    if rank == 0:
        try:
            out = fn(*args)
        except BaseException as e:      # noqa: BLE001 -- re-raised below, after the other ranks were told
            err = e
    flag = torch.tensor([0 if err is None else 1], dtype=torch.int32)
    dist.broadcast(flag, src=0)
    if err is not None:
        raise err
    if int(flag.item()):
        raise RuntimeError("rank 0 failed in %s (see its traceback)" % getattr(fn, "__name__", "a single-rank step"))
    return out


def broadcast_object(obj, src=0):
    """the object of rank `src` on every rank (small python objects: size distributions, fitted models)"""
    dist, _ = ensure_distributed()
    if dist is None:
        return obj
    box = [obj if env_rank_world()[0] == src else None]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def shared_fragment_store(bam):
    """FragmentStore of `bam`, decoded ONCE PER NODE: on every host the rank with the lowest LOCAL_RANK decodes the BAM (native
    streaming / device decoder) and publishes the per-chromosome arrays as .npy files in a private (0700, mkdtemp) directory of
    that host's shared memory (/dev/shm); the other ranks of the host map them read-only instead of decoding the whole file
    again (global pre-steps once, before sharding).  A publisher that fails tells everybody -- every
    rank raises instead of waiting for metadata that never comes --, a rank that cannot map the files decodes the BAM itself,
    and a rank that fails at that too tells everybody before anyone waits for it.  One rank: plain FragmentStore.open."""
    from .pyatac.fragments import FragmentStore
    rank, world, local = env_rank_world()
    if world <= 1 or isinstance(bam, FragmentStore):
        return FragmentStore.open(bam)
    import shutil
    import socket
    import tempfile
    dist, _ = ensure_distributed()
    host = socket.gethostname()
    who = [None] * world
    dist.all_gather_object(who, (host, local, rank))
    publisher = min((l, r) for h, l, r in who if h == host)[1]          # this host's publishing rank
    root = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
    mine, st, err, d = None, None, None, None
    if rank == publisher:
        try:
            st = FragmentStore.open(bam)
            d = tempfile.mkdtemp(prefix="natac_frags_", dir=root)           # 0700, unpredictable name
            for i, c in enumerate(st.references):
                np.save(os.path.join(d, "%d.pos.npy" % i), st.pos[c])
                np.save(os.path.join(d, "%d.tlen.npy" % i), st.tlen[c])
            mine = (host, "ok", d, st.references, st.lengths)
        except BaseException as e:      # noqa: BLE001 -- re-raised below, after the other ranks were told
            err = e
            mine = (host, "failed", "%s: %s" % (type(e).__name__, str(e)[:300]), None, None)
            if d is not None:
                shutil.rmtree(d, ignore_errors=True)
    try:
        notes = [None] * world
        dist.all_gather_object(notes, mine)
        if err is not None:
            raise err
        failed = [n for n in notes if n is not None and n[1] == "failed"]
        if failed:
            raise RuntimeError("reading %s failed on host %s (%s)" % (bam, failed[0][0], failed[0][2]))
        map_err = None
        if rank != publisher:
            _, _, pdir, refs, lens = next(n for n in notes if n is not None and n[0] == host)
            try:
                try:
                    st = FragmentStore(refs, lens, {c: np.load(os.path.join(pdir, "%d.pos.npy" % i), mmap_mode="r") for i, c in enumerate(refs)},
                                       {c: np.load(os.path.join(pdir, "%d.tlen.npy" % i), mmap_mode="r") for i, c in enumerate(refs)},
                                       trusted=True)
                    FragmentStore.register(bam, st)
                except (OSError, ValueError):                               # not visible from here after all: decode it ourselves
                    st = FragmentStore.open(bam)
            except Exception as e:      # noqa: BLE001 -- announced below: no rank may be left waiting in the gather for this one (KeyboardInterrupt /
                map_err = e             # SystemExit are NOT caught: an interrupted rank must leave at once, not enter a collective first)
        # every rank has mapped the files (or says that it could not): the publisher unlinks them (the mappings stay valid).  The
        # gather is the barrier; a rank that failed here raises its own error and every other rank raises with it.
        oks = [None] * world
        dist.all_gather_object(oks, None if map_err is None else "%s: %s" % (type(map_err).__name__, str(map_err)[:300]))
        if map_err is not None:
            raise map_err
        bad = [(r, o) for r, o in enumerate(oks) if o is not None]
        if bad:
            raise RuntimeError("rank %d could not get the fragments of %s (%s)" % (bad[0][0], bam, bad[0][1]))
    finally:
        if rank == publisher and d is not None:
            shutil.rmtree(d, ignore_errors=True)
    return st


def my_shard(packed, rank=None, world=None, kappa=4.0):
    """(lo, hi, PackedChunks subset) of this rank"""
    if rank is None or world is None:
        rank, world, _ = env_rank_world()
    lo, hi = balanced_ranges(packed.chunk_len, packed.frag_off, world, kappa)[rank]
    return lo, hi, (packed.subset(lo, hi) if hi > lo else None)


def gather_in_chunk_order(local_items, dst=0):
    """Gather per-chunk python objects / arrays from every rank to `dst`, concatenated in rank (== chunk) order.
    Works with any initialised torch.distributed backend (nccl on GPUs, gloo on CPU); no-op without one."""
    if env_rank_world()[1] <= 1 and "torch.distributed" not in __import__("sys").modules:
        return list(local_items)             # single process: do not pay the 1 s `import torch` for a no-op
    try:
        import torch.distributed as dist
    except Exception:  # pragma: no cover
        dist = None
    if dist is None or not dist.is_available() or not dist.is_initialized():
        if env_rank_world()[1] > 1:
            raise RuntimeError("WORLD_SIZE > 1 but torch.distributed is not initialised: call shard.ensure_distributed() first")
        return list(local_items)
    if dist.get_world_size() == 1:
        return list(local_items)
    world, rank = dist.get_world_size(), dist.get_rank()
    bucket = [None] * world if rank == dst else None
    dist.gather_object(list(local_items), bucket, dst=dst)
    if rank != dst:
        return None
    out = []
    for part in bucket:
        out.extend(part)
    return out


def ordered_sum(per_chunk_vectors):
    """sum per-chunk vectors strictly in chunk order (the reference's `nuc_dist += result[0]`, run_occ.py:121),
    so the floating-point result does not depend on the number of GPUs."""
    acc = None
    for v in per_chunk_vectors:
        acc = np.array(v, dtype=np.float64) if acc is None else acc + v
    return acc
