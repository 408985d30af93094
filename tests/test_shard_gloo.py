"""CPU, world_size 2 over gloo: the N>1 path (contiguous balanced shards, rank-local processing, ordered gather)
gives exactly the single-process result.  The per-chunk work is done by the CPU oracle here (no GPU in this tier)."""
import os
import socket

import numpy as np
import pytest

from nucleoatac_amd.shard import balanced_ranges, gather_in_chunk_order, my_shard, ordered_sum
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _chunk_result(pk, k, nucp, nfrp, gk=None):
    gk = k if gk is None else gk
    from oracle import natac_oracle as O
    l, n = pk.chunk_frags(k)
    L = int(pk.chunk_len[k])
    oc = O.occ_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, L, pk.chunk_bias(k), -pk.bias_left, nucp, nfrp)
    ins = O.get_insertions(l.astype(np.int64), n.astype(np.int64), 0, L)
    dist = np.sum(oc["mat"][:, 60:60 + 121], axis=1) * (1.0 / (gk + 3))
    return dict(occ_sum=float(np.nansum(oc["smoothed_vals"])), ins=ins.astype(np.int32), nuc_dist=dist)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pk = make_synthetic_chunks(6, 420, 60, seed=9, poisson=True)
    nucp, nfrp = synth_occ_distributions(251)
    lo, hi, sub = my_shard(pk, rank, world)
    local = [_chunk_result(sub, k, nucp, nfrp, lo + k) for k in range(hi - lo)] if sub is not None else []
    allr = gather_in_chunk_order(local, dst=0)
    if rank == 0:
        q.put((lo, hi, allr))
    dist.barrier()
    dist.destroy_process_group()


def test_balanced_ranges_cover_and_balance():
    pk = make_synthetic_chunks(1000, 700, 100, seed=2, poisson=True)
    for world in (1, 2, 3, 8):
        r = balanced_ranges(pk.chunk_len, pk.frag_off, world)
        assert r[0][0] == 0 and r[-1][1] == pk.n_chunks
        assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
        cost = [(pk.chunk_len[a:b].sum() + 4.0 * (pk.frag_off[b] - pk.frag_off[a])) for a, b in r]
        assert max(cost) <= 1.05 * (sum(cost) / world) + 2000
    # more ranks than chunks: empty shards are allowed
    r = balanced_ranges([100, 100], [0, 5, 10], 4)
    assert r[0][0] == 0 and r[-1][1] == 2 and sum(b - a for a, b in r) == 2


def test_two_rank_gloo_equals_single_process():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    lo, hi, allr = q.get(timeout=300)
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    pk = make_synthetic_chunks(6, 420, 60, seed=9, poisson=True)
    nucp, nfrp = synth_occ_distributions(251)
    ref = [_chunk_result(pk, k, nucp, nfrp) for k in range(pk.n_chunks)]
    assert lo == 0 and 0 < hi < pk.n_chunks and len(allr) == pk.n_chunks
    for a, b in zip(allr, ref):
        assert a["occ_sum"] == b["occ_sum"] and np.array_equal(a["ins"], b["ins"])
    # the cross-chunk reduction is done in chunk order -> bit-identical to the unsharded sum
    assert np.array_equal(ordered_sum([a["nuc_dist"] for a in allr]), ordered_sum([b["nuc_dist"] for b in ref]))


def test_bench_cfg4_shards_partition_the_workload():
    """bench.py --workload cfg4: every rank draws the same count vector, balances the chunk list and generates only its shard
    from counter-seeded blocks; the shards of 1, 2 and 3 ranks are the same chunks (fragments, bias, coordinates) in the same
    order, whatever the sub-batch size"""
    import argparse
    import importlib.util
    import os
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    def shards(world, sub):
        a = argparse.Namespace(workload="cfg4", chunks=5000, chunk_len=700, frags_per_chunk=60, sub_chunks=sub, seed=0)
        out = []
        for r in range(world):
            subs, desc, info = bench.make_workload(a, r, world)
            out.append((subs, info))
        return out

    def flat(parts):
        pks = [pk for subs, _ in parts for pk in subs]
        return (np.concatenate([p.chunk_start for p in pks]), np.concatenate([p.chunk_len for p in pks]),
                np.concatenate([np.diff(p.frag_off) for p in pks]), np.concatenate([p.frag_lpos for p in pks]),
                np.concatenate([p.frag_ilen for p in pks]), np.concatenate([p.bias_log for p in pks]))

    one = flat(shards(1, 1700))
    assert len(one[0]) == 5000 and np.all(np.diff(one[0]) > 0)
    for world, sub in ((2, 1700), (3, 900), (2, 2500)):
        parts = shards(world, sub)
        got = flat(parts)
        for x, y in zip(one, got):
            assert np.array_equal(x, y), (world, sub)
        imb = parts[0][1]["imbalance"]
        assert len(imb["bp_per_rank"]) == world and imb["bp_max_over_mean"] < 1.02 and parts[0][1]["scaling"] == "strong"


def _control_plane_worker(rank, world, port, q, bam_ok, bam_bad):
    """ensure_distributed(prefer="nccl") without GPUs, then shared_fragment_store: publish / map, and a failing publisher"""
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nucleoatac_amd import shard
    t0 = time.time()
    dist, created = shard.ensure_distributed(prefer="nccl")
    t_init = time.time() - t0
    backend = shard.control_backend()
    shard.barrier(sync_cuda=True)                       # the bench's barrier form must work on the fallback too
    st = shard.shared_fragment_store(bam_ok)
    reads = {c: (np.array(st.pos[c]), np.array(st.tlen[c])) for c in st.references}
    mapped = any(isinstance(st.pos[c], np.memmap) for c in st.references)
    t0 = time.time()
    try:
        shard.shared_fragment_store(bam_bad)
        failed = None
    except Exception as e:      # noqa: BLE001
        failed = "%s: %s" % (type(e).__name__, e)
    t_fail = time.time() - t0
    # a NON-publisher that can neither map nor decode: every rank raises together (nobody waits in a barrier for it)
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    if rank == 1:
        def broken(*a, **k):
            raise RuntimeError("mapping broke on this rank")
        FragmentStore.register = staticmethod(broken)
    t0 = time.time()
    try:
        shard.shared_fragment_store(bam_ok + ".copy.npz")
        failed2 = None
    except Exception as e:      # noqa: BLE001
        failed2 = "%s: %s" % (type(e).__name__, e)
    failed = (failed, failed2, time.time() - t0)
    shard.barrier()                 # the publisher removes its directory right after the gather: look only once every rank is past it
    left = [f for f in os.listdir("/dev/shm") if f.startswith("natac_frags_")] if os.path.isdir("/dev/shm") else []
    shard.barrier()
    q.put((rank, created, backend, t_init, reads, mapped, failed, t_fail, left))
    dist.destroy_process_group()


def test_control_plane_without_gpus_and_fragments_published_per_node(tmp_path):
    """two gloo ranks, no GPU: asking for RCCL lands on gloo within seconds on both ranks (decided collectively); the reads of a
    BAM stand-in are decoded by the node's first rank and mapped by the other; a publisher that fails raises on BOTH ranks at
    once (nobody waits for metadata), and nothing is left in /dev/shm"""
    import torch.multiprocessing as mp
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    rng = np.random.default_rng(3)
    pos = {"chrA": np.sort(rng.integers(0, 100000, 5000)), "chrB": np.sort(rng.integers(0, 50000, 1200))}
    tl = {c: rng.integers(30, 400, len(p)) for c, p in pos.items()}
    bam_ok = str(tmp_path / "reads.bam.npz")
    FragmentStore(["chrA", "chrB"], [100000, 50000], pos, tl).save_npz(bam_ok)
    FragmentStore(["chrA", "chrB"], [100000, 50000], pos, tl).save_npz(bam_ok + ".copy.npz")
    bam_bad = str(tmp_path / "missing.bam.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_control_plane_worker, args=(r, 2, port, q, bam_ok, bam_bad)) for r in range(2)]
    for p in procs:
        p.start()
    rows = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    ref = FragmentStore.open(bam_ok)
    for rank, created, backend, t_init, reads, mapped, failed, t_fail, left in rows:
        assert created and backend == "gloo" and t_init < 60
        assert mapped == (rank == 1)                     # rank 0 published, rank 1 mapped the shared arrays
        for c in ref.references:
            assert np.array_equal(reads[c][0], ref.pos[c]) and np.array_equal(reads[c][1], ref.tlen[c])
        assert failed[0] is not None and t_fail < 60, (failed, t_fail)
        assert failed[1] is not None and "mapping broke" in failed[1] and failed[2] < 60, failed      # on BOTH ranks, at once
        assert not left
