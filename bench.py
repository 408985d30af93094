#!/usr/bin/env python3
"""bench.py -- Mbp/s through the occ + nuc signal pipeline on N MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over one batch of synthetic chunks that is already
resident in HBM: nuc tracks (coverage, raw, background, norm, smooth) + occupancy tracks (grid MLE,
smoothing, cov, NaN fill) + per-base insertion counts + device-side candidate search (call_peaks) with per-candidate
LR / variance / z.
Workload at N=1 = BASELINE.json configs[2]: synthetic 100k windows x 2 kb (2,120 bp after the +-60
slop), 50 M fragments, default VMat (146 x 121).  With N > 1 every rank owns its own shard of the
same shape (chunk list sharded across GPUs, no data-path collective): weak scaling.

Launch: python bench.py --gpus 1            (default)
        python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_BP = 79.8          # SURVEY.md section 8(d), config 3 (compulsory HBM traffic of the whole path)
FLOP_PER_BP_BG = 2 * 146 * 121   # fp64 flop per base of the background correlation evaluated directly (R x W FMA)
# executed by the FFT kernel: 73 row pairs x 364 flop per lane (172 add + 72 mul + 60 fma) x 64 lanes per 392-base tile
FLOP_PER_BP_BG_FFT = 73 * 364 * 64 / 392.0
KERNEL_LABEL = {"background": "natac_background_fft (dense bias x VMat correlation, fp64 FFT)",
                "occ_mle": "natac_occ_mle<5,60,0,1> (occupancy grid MLE)",
                "candidates": "natac_candidates4 + peak search (LR / variance / z of the candidates)"}
KERNEL_SYMBOL = {"background": "natac_background_fft", "occ_mle": "natac_occ_mle", "candidates": "natac_candidates4"}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6          # MI355X fp64 vector peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunks", type=int, default=100000, help="chunks per GPU")
    ap.add_argument("--chunk-len", type=int, default=2120)
    ap.add_argument("--frags-per-chunk", type=int, default=500)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks in the CPU baseline's optimised-mode sample (0 = 2000)")
    ap.add_argument("--cpu-literal-chunks", type=int, default=0, help="chunks in the literal-mode sample (0 = auto)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend (nccl == RCCL on ROCm)")
    ap.add_argument("--share-device", action="store_true",
                    help="functional test of the N>1 path on a 1-GPU box: every rank uses GPU 0 (use with --dist-backend gloo)")
    return ap.parse_args()


def _cpu_chunk(args):
    """one chunk through the CPU oracle in the reference's execution shape: what _occHelper + _nucHelper do per chunk
    (nucleoatac/run_occ.py:23-39, run_nuc.py:22-39): OccChunk.process (grid MLE, smoothing, coverage, callPeaks) and
    NucChunk.process up to findAllNucs (tracks, call_peaks, getLR for nuc_cov > min_reads, calculateCov / z for
    lr > min_lr, NucleosomeCalling.py:294-315) + the insertion track.  literal: dense scipy.signal.correlate and the O(N^2)
    calculateCov pair sum like the reference; otherwise per-row correlations and the closed-form variance."""
    from scipy import signal
    from oracle import natac_oracle as O
    (l, n, L, bias, bl, vmat, vlo, vup, sizes, nucp, nfrp, literal) = args
    l = l.astype(np.int64)
    n = n.astype(np.int64)
    kw = {}
    if literal:
        kw["dense_correlate"] = lambda sub, vm: signal.correlate(sub, vm, mode="valid")[0]   # NucleosomeCalling.py:34
    nt = O.nuc_chunk_tracks(l, n, 0, L, bias, -bl, vmat, vlo, vup, sizes, **kw)
    comb = nt["norm"] + nt["smoothed"]
    cands = O.call_peaks(comb.copy(), min_signal=0, sep=25, boundary=60, order=12)
    w = vmat.shape[1] // 2
    nz = 0
    acc = 0.0
    for p in cands:
        p = int(p)
        if nt["nuc_cov"][p] > 1:                                           # min_reads = 1 (NucleosomeCalling.py:210, 304)
            lr = O.get_lr(nt["mat"], nt["mat_start"], nt["bmat"], nt["b0"], nt["b_start"], vmat, vlo, vup, p)
            if lr > 0:                                                     # min_lr = 0 (:306)
                pr = O.signal_distribution_probs(nt["bmat"], nt["b_start"], vlo, vup, w, p)
                z, _ = O.z_score(nt["norm"][p], nt["nuc_cov"][p], pr, vmat, literal=literal)
                acc += z
                nz += 1
    oc = O.occ_chunk_tracks(l, n, 0, L, bias, -bl, nucp, nfrp)
    O.call_peaks(oc["smoothed_vals"], sep=120, min_signal=0.1)              # OccChunk.callPeaks (Occupancy.py:225-231)
    O.get_insertions(l, n, 0, L)
    return (float(np.nansum(nt["norm"]) + np.nansum(oc["smoothed_vals"]) + acc), len(cands), nz)


def _cpu_worker_init():
    """the reference's workers are single-threaded numpy processes: pin BLAS/OpenMP pools to one thread each"""
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(1)
    except Exception:
        pass


def _cpu_mode(pool, workers, cores, pk, par, sizes, nucp, nfrp, n_chunks, literal):
    """time n_chunks chunks in the reference's shape: `chunks.split(items = cores*5)` rounds, one pool.map per round
    (run_occ.py:101-123, run_nuc.py:164-188)"""
    tasks = []
    for k in range(n_chunks):
        l, n = pk.chunk_frags(k)
        tasks.append((l, n, int(pk.chunk_len[k]), pk.chunk_bias(k), pk.bias_left, par["vmat"], int(par["vlower"]),
                      int(par["vupper"]), sizes, nucp, nfrp, literal))
    bp = int(pk.chunk_len[:n_chunks].sum())
    per_round = cores * 5
    ncand = nz = 0
    t0 = time.time()
    for r0 in range(0, n_chunks, per_round):
        for _, c, z in pool.map(_cpu_chunk, tasks[r0:r0 + per_round]):
            ncand += c
            nz += z
    dt = time.time() - t0
    return dict(value=round(bp / dt / 1e6, 5), unit="Mbp/s", chunks=n_chunks, bp=bp, seconds=round(dt, 1),
                candidates=ncand, z_scores=nz, rounds_of=per_round)


def cpu_baseline(pk, par, sizes, nucp, nfrp, n_chunks, n_literal):
    """the CPU restatement run in the reference's execution shape -- multiprocessing.Pool(cores-1),
    chunks.split(cores*5) rounds -- on the host cores of this box, on a bounded sample of the same chunks.
    "literal" = dense correlate + O(N^2) calculateCov as the reference executes them (the reported `value`);
    "optimised" = per-row correlations + closed-form variance."""
    import multiprocessing as mp
    cores = _effective_cores()
    workers = max(1, cores - 1)
    n_chunks = min(n_chunks if n_chunks > 0 else 2000, pk.n_chunks)
    n_literal = min(n_literal if n_literal > 0 else max(5 * cores, 160), pk.n_chunks)
    ctx = mp.get_context("fork")
    with ctx.Pool(workers, initializer=_cpu_worker_init) as pool:
        pool.map(_cpu_chunk, [(pk.chunk_frags(0)[0], pk.chunk_frags(0)[1], int(pk.chunk_len[0]), pk.chunk_bias(0), pk.bias_left,
                               par["vmat"], int(par["vlower"]), int(par["vupper"]), sizes, nucp, nfrp, False)] * workers)  # warm-up (imports)
        opt = _cpu_mode(pool, workers, cores, pk, par, sizes, nucp, nfrp, n_chunks, False)
        lit = _cpu_mode(pool, workers, cores, pk, par, sizes, nucp, nfrp, n_literal, True)
    return dict(value=lit["value"], unit="Mbp/s", cores=workers, kind="port",
                sample="literal mode (scipy dense correlate + O(N^2) calculateCov, as the reference runs): %d of the workload's "
                       "chunks, %d bp, %.1f s; optimised mode (per-row correlate, closed-form variance): %d chunks, %d bp, %.1f s; "
                       "oracle/natac_oracle.py: occ (MLE + smoothing + peaks) + nuc (tracks + call_peaks + LR + z) + ins per chunk, "
                       "Pool(%d) in rounds of cores*5 = %d chunks" % (lit["chunks"], lit["bp"], lit["seconds"], opt["chunks"],
                                                                      opt["bp"], opt["seconds"], workers, cores * 5),
                literal=lit, optimised=opt, host_cores_visible=len(os.sched_getaffinity(0)), cores_by_cgroup_quota=cores)


def _effective_cores():
    """CPUs this process can really use: the visible ones capped by the cgroup CPU quota (cpu.max "1600000 100000" = 16)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(p))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // p)))
        except (OSError, ValueError):
            pass
    return n


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC summary (profiles/*/pmc_summary.csv):
    (2 x FETCH_SIZE + WRITE_SIZE) x 1024 -- FETCH_SIZE under-counts streaming reads 2x on gfx950
    (MI355X_MICROARCH.md, HBM section).  None when no profile is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.csv")))
    if not files:
        return None
    f = w = None
    with open(files[-1]) as fh:
        for row in csv.reader(l for l in fh if not l.startswith("#")):
            if len(row) == 4 and kernel_substr in row[0]:
                if row[1] == "FETCH_SIZE":
                    f = float(row[2]) / float(row[3])
                if row[1] == "WRITE_SIZE":
                    w = float(row[2]) / float(row[3])
    if f is None or w is None:
        return None
    return (2.0 * f + w) * 1024.0


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if a.share_device:
        local_rank = 0
    if world > 1:
        import torch
        import torch.distributed as dist
        if a.dist_backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=a.dist_backend)

    from nucleoatac_amd import _lib as L
    from nucleoatac_amd.device import Context
    from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution

    par = np.load(os.path.join(ROOT, "tests", "golden", "params_example.npz"))
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    t_gen = time.time()
    pk = make_synthetic_chunks(a.chunks, a.chunk_len, a.frags_per_chunk, seed=a.seed + 1000 * rank)
    t_gen = time.time() - t_gen
    # CPU baseline first (rank 0, N=1 only): it forks worker processes, so run it before the HIP context exists
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(pk, par, sizes, nucp, nfrp, a.cpu_chunks, a.cpu_literal_chunks)

    ctx = Context(local_rank)
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(sizes)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    t_up = time.time()
    batch = ctx.upload(pk)
    ctx.sync()
    t_up = time.time() - t_up

    n_cand = [0]

    def step():
        batch.run_nuc(10)
        batch.run_occ()
        batch.run_ins(0, 2000)
        # candidate search (call_peaks, sep 25 / order 12 / boundary 60 as NucChunk.findAllNucs) + LR / var / z, on the device
        # like the per-base tracks, the candidate arrays stay resident in HBM inside the timed region
        n_cand[0] = batch.run_peaks(min_signal=0, sep=25, boundary=60, order=12, download=False)

    on_gpu = dist is not None and a.dist_backend == "nccl"

    def barrier():
        ctx.sync()
        if dist is not None:
            if on_gpu:
                import torch
                torch.cuda.synchronize()
            dist.barrier()

    for _ in range(a.warmup):
        step()
    ctx.profile_enable(True)
    ctx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    ctx.sync()
    if on_gpu:
        import torch
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    barrier()
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if on_gpu else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = ctx.profile()
    total_bp = pk.total_bp * world
    ms_per_step = dt / a.steps * 1e3
    value = total_bp * a.steps / dt / 1e6

    # PCIe-inclusive figure (never `value`): one upload + one download of the per-base tracks that the writers consume
    t_dn = time.time()
    for t in (L.T_NORM, L.T_SMOOTH, L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER):
        batch.track(t)
    cand = batch.download_peaks(n_cand[0])
    t_dn = time.time() - t_dn
    assert len(cand[0]) == n_cand[0] and np.isfinite(cand[4][:1000]).any()

    if rank == 0:
        # roofline of the dominant kernel class of this run (largest HIP-event time on the launch stream)
        dom = max(("background", "occ_mle", "candidates"), key=lambda k: prof[k][0])
        dom_ms, dom_n = prof[dom]
        dom_avg_s = (dom_ms / max(1, dom_n)) / 1e3
        traffic = pmc_traffic_bytes(KERNEL_SYMBOL[dom])
        alg_bytes = ALG_BYTES_PER_BP * pk.total_bp
        achieved = alg_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
        bg_ms, bg_n = prof["background"]
        bg_avg_s = (bg_ms / max(1, bg_n)) / 1e3
        direct_tflops = FLOP_PER_BP_BG * pk.total_bp / bg_avg_s / 1e12 if bg_avg_s > 0 else 0.0
        fft_tflops = FLOP_PER_BP_BG_FFT * pk.total_bp / bg_avg_s / 1e12 if bg_avg_s > 0 else 0.0
        out = {
            "metric": "Mbp/s through occ+nuc signal pipeline", "value": round(value, 3), "unit": "Mbp/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[2]: synthetic %d windows x 2 kb (L=%d after slop), %d fragments, default VMat "
                                   "146x121, 1 GPU-shard per rank" % (a.chunks, a.chunk_len, pk.n_frags),
                       "chunks_per_gpu": a.chunks, "chunk_len": a.chunk_len, "fragments_per_gpu": pk.n_frags,
                       "candidates_per_gpu": int(n_cand[0]), "sharding": "chunk list split across ranks, no collective"},
            "roofline": {"bound": "hbm", "kernel": KERNEL_LABEL[dom],
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "avg_launch_ms": round(dom_avg_s * 1e3, 3), "launches": int(dom_n),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "note": "the path is fp64-VALU / LDS bound, not HBM bound (SURVEY 8d: ~5e4 flop/base against 79.8 B/base); "
                                 "background_fp64 gives the arithmetic rate of the background kernel",
                         "background_fp64": {"avg_launch_ms": round(bg_avg_s * 1e3, 3),
                                             "direct_equivalent_tflops": round(direct_tflops, 2),
                                             "executed_tflops": round(fft_tflops, 2), "peak": FP64_PEAK_TFLOPS,
                                             "unit": "TFLOP/s", "frac_executed": round(fft_tflops / FP64_PEAK_TFLOPS, 4)}},
            "kernels_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in prof.items()},
            "host": {"generate_s": round(t_gen, 2), "upload_s": round(t_up, 2), "download_5_tracks_and_candidates_s": round(t_dn, 2),
                     "pcie_inclusive_mbp_s": round(pk.total_bp / (dt / a.steps + t_up + t_dn) / 1e6, 2)},
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out))
    batch.free()
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
