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


def ensure_distributed():
    """Under torchrun (WORLD_SIZE > 1) make sure the process group exists before any rank-dependent work: run_occ /
    run_nuc called through the API (not cli.main) would otherwise skip the barrier and the gather silently and rank 0
    would merge part files that other ranks are still writing.  Backend: RCCL ("nccl") or NATAC_DIST_BACKEND=gloo.
    Returns (dist module or None, True if this call created the group)."""
    rank, world, local = env_rank_world()
    if world <= 1:
        return None, False
    import torch.distributed as dist
    if dist.is_initialized():
        if dist.get_world_size() != world:
            raise RuntimeError("WORLD_SIZE=%d but the torch.distributed group has %d ranks" % (world, dist.get_world_size()))
        return dist, False
    backend = os.environ.get("NATAC_DIST_BACKEND", "nccl")
    if backend == "nccl":
        # only barriers and the gather of small per-chunk results go through the group (never track data): if RCCL cannot come
        # up on this node the run continues over gloo
        import sys
        import torch
        dev = int(os.environ.get("NATAC_DEVICE", local))          # the GPU this rank computes on (default: LOCAL_RANK)
        try:
            torch.cuda.set_device(dev)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
            probe = torch.ones(1, device="cuda")
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            if int(probe.item()) != world:
                raise RuntimeError("all_reduce probe returned %r" % probe.item())
        except Exception as e:      # noqa: BLE001
            sys.stderr.write("nucleoatac_amd: RCCL unavailable (%s: %s); process group over gloo\n" % (type(e).__name__, str(e)[:200]))
            try:
                dist.destroy_process_group()
            except Exception:       # noqa: BLE001
                pass
            os.environ["MASTER_PORT"] = str(int(os.environ.get("MASTER_PORT", "29500")) + 1)
            os.environ["TORCHELASTIC_USE_AGENT_STORE"] = "False"      # rank 0 hosts the new store itself
            dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend=backend)
    return dist, True


def barrier():
    """all ranks reach this point (no-op on one rank; raises if WORLD_SIZE > 1 without a process group)"""
    dist, _ = ensure_distributed()
    if dist is not None:
        dist.barrier()


def my_shard(packed, rank=None, world=None, kappa=4.0):
    """(lo, hi, PackedChunks subset) of this rank"""
    if rank is None or world is None:
        rank, world, _ = env_rank_world()
    lo, hi = balanced_ranges(packed.chunk_len, packed.frag_off, world, kappa)[rank]
    return lo, hi, (packed.subset(lo, hi) if hi > lo else None)


def gather_in_chunk_order(local_items, dst=0):
    """Gather per-chunk python objects / arrays from every rank to `dst`, concatenated in rank (== chunk) order.
    Works with any initialised torch.distributed backend (nccl on GPUs, gloo on CPU); no-op without one."""
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
