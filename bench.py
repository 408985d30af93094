#!/usr/bin/env python3
"""bench.py -- Mbp/s through the occ + nuc signal pipeline on N MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over the rank's synthetic chunks, inputs already resident in HBM:
nuc tracks (coverage, raw, background, norm, smooth) + occupancy tracks (grid MLE, smoothing, cov, NaN fill) + per-base
insertion counts + device-side candidate search (call_peaks) with per-candidate LR / variance / z.

Workloads (--workload):
  cfg3 (default)  BASELINE.json configs[2]: 100k windows x 2 kb (2,120 bp after the +-60 slop), 50 M fragments, default
                  VMat (146 x 121).  With N > 1 every rank owns its own shard of the same shape (weak scaling).
  cfg3-heavy      the same windows with Poisson fragment counts and 1 % of the chunks 10x denser (heavy-tailed load).
  cfg5            BASELINE.json configs[4]: 8 independent samples, one per GPU (every rank its own 12,500-chunk configs[2]-like set
                  with its own seed: replicas, weak scaling); after the timed steps every rank runs the multinomial_cov tolerance
                  sweep (literal O(N^2) fp64 / closed form fp64 / closed form fp32 at the candidates of 200 chunks) and rank 0
                  reports the worst relative errors over all samples.
  cfg4            BASELINE.json configs[3]: ~300k tiles x 10 kb (10,120 bp), ~200 M fragments, the chunk list sharded across
                  the N ranks by sum(L) + kappa sum(F) (nucleoatac_amd/shard.py): STRONG scaling, total work fixed.  Every
                  rank draws the same cheap per-chunk count vector, balances, and generates only its own shard in
                  counter-seeded blocks; a shard is processed in sub-batches whose outputs are recycled
                  (natac_batch_release_outputs) when they do not all fit in HBM.

Also reported (rank 0, N = 1): the host-to-host rate -- packed inputs in (pinned) host memory -> per-base tracks back in
(pinned) host memory, sub-batches pipelined over six contexts so that uploads, kernels and downloads overlap -- and the CPU
baseline (the oracle in the reference's Pool.map shape on the box's host cores).

Launch: python bench.py --gpus N            (N = 1 by default)
`--gpus N` MEANS N ranks, one per GPU, whoever launches it: started plainly with N > 1 (no WORLD_SIZE in the environment) the script
starts its own N ranks through `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on 127.0.0.1 and a free port and
hands their single JSON line through (like the reference's `--cores N`, which forks its own pool: nucleoatac/run_occ.py:101-102);
started by a launcher that already set WORLD_SIZE, it insists on WORLD_SIZE == --gpus and exits with status 2 otherwise.
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

ALG_BYTES_PER_BP = {"cfg3": 79.8, "cfg3-heavy": 79.8, "cfg4": 76.9, "cfg5": 79.8}   # 76 L + 8 F + 3,944 B per chunk
FLOP_PER_BP_BG = 2 * 146 * 121   # fp64 flop per base of the background correlation evaluated directly (R x W FMA)
# executed by the FFT kernel: 73 row pairs x 364 flop per lane (172 add + 72 mul + 60 fma as of round 4; round 5 folds the 1/sqrt2
# of dft8 into FMAs -- 132 add + 44 mul + 100 fma, 276 instructions instead of 304 for the same arithmetic -- and keeps this count)
# x 64 lanes per 392-base tile
FLOP_PER_BP_BG_FFT = 73 * 364 * 64 / 392.0     # plain tiles; a run's own figure comes from its chunks' tilings (bg_executed_flop)
FLOP_PER_TILE_BG_FFT = 73 * 364 * 64
# the edge pass of an extended tile (natac_fft_bg.hpp, bg_edge_side): per side 37 steps of one 16 x 16 x 4 fp64 MFMA (2,048 flop, half of
# the 16 x 16 products are outputs nobody reads: executed, not useful) + per lane 2 multiplications, 1 subtraction, 1 FMA
FLOP_PER_TILE_BG_EDGE = 2 * 37 * (2048 + 64 * 5)


def bg_executed_flop(ctx, subs):
    """fp64 operations the background stage executes on these chunks: tiles and extended tiles as the library cuts them (natac_bg_tiling)"""
    flop, tiles, ext_tiles = 0, 0, 0
    for sub in subs:
        lens, counts = np.unique(np.asarray(sub.chunk_len), return_counts=True)
        for Lc, k in zip(lens.tolist(), counts.tolist()):
            nt, ex = ctx.bg_tiling(int(Lc))
            tiles += nt * k
            ext_tiles += ex * k
    flop = tiles * FLOP_PER_TILE_BG_FFT + ext_tiles * FLOP_PER_TILE_BG_EDGE
    return flop, tiles, ext_tiles
KERNEL_LABEL = {"background": "natac_background_fft (dense bias x VMat correlation, fp64 FFT)",
                "occ_mle": "natac_occ_gsum + natac_occ_decide (occupancy grid MLE)",
                "candidates": "natac_peaks_chunk_reg + natac_candidates_paired (candidate search; LR / variance / z)"}
KERNEL_SYMBOL = {"background": "natac_background_fft", "occ_mle": "natac_occ_", "candidates": "natac_candidates_paired"}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8 TB/s spec
FP64_PEAK_TFLOPS = 78.6          # MI355X fp64 vector peak (256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
H2H_TRACKS = ("T_NORM", "T_SMOOTH", "T_OCC", "T_OCC_LOWER", "T_OCC_UPPER")   # what `nucleoatac run` writes by default


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None,
                    help="number of ranks = GPUs (default: WORLD_SIZE if a launcher set it, else 1); N > 1 without a launcher: the script "
                         "starts its own N ranks")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", choices=["cfg3", "cfg3-heavy", "cfg4", "cfg5"], default="cfg3")
    ap.add_argument("--chunks", type=int, default=0, help="chunks per GPU (cfg3: 100000) / in total (cfg4: 300000)")
    ap.add_argument("--chunk-len", type=int, default=0)
    ap.add_argument("--frags-per-chunk", type=int, default=0)
    ap.add_argument("--sub-chunks", type=int, default=20000, help="cfg4: chunks per sub-batch")
    ap.add_argument("--recycle", choices=["auto", "on", "off"], default="auto",
                    help="release every sub-batch's outputs after its stages (auto: when they do not all fit in HBM)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-chunks", type=int, default=0, help="chunks in the CPU baseline's optimised-mode sample (0 = 2000)")
    ap.add_argument("--cpu-literal-chunks", type=int, default=0, help="chunks in the literal-mode sample (0 = auto)")
    ap.add_argument("--no-h2h", action="store_true", help="skip the pipelined host-to-host measurement")
    ap.add_argument("--h2h-sub", type=int, default=2500, help="chunks per pipelined sub-batch")
    ap.add_argument("--h2h-threads", type=int, default=6, help="contexts (host threads) of the host-to-host pipeline")
    ap.add_argument("--h2h-text-sub", type=int, default=0, help="chunks per sub-batch of the bedGraph.gz form (0: 5000, or --h2h-sub if that is given)")
    ap.add_argument("--h2h-text-threads", type=int, default=0, help="contexts of the bedGraph.gz form (0: 10, or --h2h-threads if that is given)")
    ap.add_argument("--h2h-ranks", action="store_true", help="N > 1: every rank also runs the host-to-host pipeline on (a part of) its shard, "
                    "all ranks at once; the rates are reported per rank")
    ap.add_argument("--h2h-rank-chunks", type=int, default=20000, help="chunks of a rank's shard the --h2h-ranks leg runs on")
    ap.add_argument("--cli-chunks", type=int, default=10000, help="chunks of the CLI end-to-end measurement (0 = skip)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dist-backend", default="gloo", choices=["gloo", "nccl"],
                    help="barrier backend.  The path has no collective (chunks are independent), so the default control plane is gloo + a "
                         "device synchronise per rank; nccl additionally forms an RCCL group (probed, used only if every rank has it)")
    ap.add_argument("--share-device", action="store_true",
                    help="functional test of the N>1 path on a 1-GPU box: every rank uses GPU 0 (use with --dist-backend gloo)")
    return ap.parse_args()


def resolve_ranks(gpus, environ):
    """what `--gpus` and the launcher's environment say together: ("run", world) = this process is one of `world` ranks (or the only one);
    ("spawn", n) = started plainly with --gpus n > 1, the script has to start its own n ranks; ("error", message) = a launcher's
    WORLD_SIZE contradicts --gpus (a line that says n_gpus = X must have been measured on X ranks)."""
    ws = environ.get("WORLD_SIZE")
    if ws is None:
        if gpus is None or gpus == 1:
            return "run", 1
        if gpus < 1:
            return "error", "bench.py: --gpus %d" % gpus
        return "spawn", gpus
    world = int(ws)
    if gpus is not None and gpus != world:
        return "error", ("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks; start `python bench.py --gpus %d` plainly (it launches "
                         "its own ranks) or give the launcher --nproc-per-node %d" % (gpus, world, gpus, gpus))
    return "run", world


def spawn_ranks(n):
    """re-exec this command line under torch.distributed.run with n ranks on this node (rendezvous on 127.0.0.1, a free port); the
    children inherit stdout / stderr, so rank 0's JSON line is this process's line"""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ------------------------------------------------------------------------------------------------ CPU baseline
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


def _cpu_mode(pool, cores, pk, par, sizes, nucp, nfrp, n_chunks, literal):
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
                               par["vmat"], int(par["vlower"]), int(par["vupper"]), sizes, nucp, nfrp, False)] * workers)  # warm-up
        opt = _cpu_mode(pool, cores, pk, par, sizes, nucp, nfrp, n_chunks, False)
        lit = _cpu_mode(pool, cores, pk, par, sizes, nucp, nfrp, n_literal, True)
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


def pmc_source():
    """where the committed counter numbers come from: newest profiles/*/pmc_summary.csv, the source stamp it was collected
    with (tools/pmc_summarize.py) and whether that stamp matches the sources of this run"""
    import glob
    from nucleoatac_amd._lib import profile_sha16
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.csv")))
    if not files:
        return None
    stamp = None
    with open(files[-1]) as fh:
        for l in fh:
            if l.startswith("# source_sha16="):
                stamp = l.split("=", 1)[1].split()[0]
    cur = profile_sha16()
    return {"file": os.path.relpath(files[-1], ROOT), "collected_at_source_sha16": stamp, "current_source_sha16": cur,
            "stale": stamp != cur,
            "note": "counters are read from the committed rocprofv3 --pmc summary, not measured in this run; the stamp covers the library's "
                    "sources, nucleoatac_amd/synth.py and bench.py's workload builder"}


def pmc_traffic_bytes(kernel_substr):
    """HBM bytes per step of the kernels whose name contains `kernel_substr`, from the newest committed rocprofv3 PMC summary
    (profiles/*/pmc_summary.csv): (FETCH_SIZE + WRITE_SIZE) x 1024.  FETCH_SIZE is NOT doubled: the guide's x2 correction
    is calibrated for 16-byte-per-lane streaming reads only; these kernels read 8 bytes per lane (uncalibrated, so the raw
    counter is reported).  None when no profile is committed."""
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
                    f = (f or 0.0) + float(row[2]) / float(row[3])
                if row[1] == "WRITE_SIZE":
                    w = (w or 0.0) + float(row[2]) / float(row[3])
    if f is None or w is None:
        return None
    return (f + w) * 1024.0


def pmc_step_traffic():
    """HBM bytes of ONE STEP over ALL kernels of the newest committed PMC summary: sum of FETCH_SIZE and WRITE_SIZE (KiB) of the last
    dispatch of every kernel -- raw, and with the guide's x2 on FETCH_SIZE (MI355X_MICROARCH.md: gfx950 under-counts wide streaming
    reads 2x; an upper bound for kernels that read 8 bytes per lane)"""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.csv")))
    if not files:
        return None
    f = w = 0.0
    with open(files[-1]) as fh:
        for row in csv.reader(l for l in fh if not l.startswith("#")):
            if len(row) == 4 and row[1] in ("FETCH_SIZE", "WRITE_SIZE") and row[0].lstrip('"').startswith(("natac::", "void natac::")):
                v = float(row[2]) / float(row[3]) * 1024.0
                if row[1] == "FETCH_SIZE":
                    f += v
                else:
                    w += v
    return dict(fetch_bytes=f, write_bytes=w) if (f or w) else None


def reference_calibration():
    """reference vs port seconds per chunk on identical inputs (tests/golden/calibrate_cpu_baseline.py, build container)"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "cpu_baseline_calibration.json")))
    if not files:
        return None
    with open(files[-1]) as fh:
        d = json.load(fh)
    d["file"] = os.path.relpath(files[-1], ROOT)
    return d


def pmc_valu_issue():
    """{kernel: share of the SIMDs' issue cycles spent on VALU instructions} for the heavy kernels, from the newest committed PMC
    summary: SQ_INSTS_VALU (wave-instructions) x 4 cycles (a wave64 fp64 / fp32 instruction occupies its SIMD for 4 cycles) over
    1,024 SIMDs x the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs).  None when no profile is committed."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.csv")))
    if not files:
        return None
    per = {}
    with open(files[-1]) as fh:
        for row in csv.reader(l for l in fh if not l.startswith("#")):
            if len(row) == 4 and row[1] in ("SQ_INSTS_VALU", "GRBM_GUI_ACTIVE", "kernel_ns_under_pmc"):
                per.setdefault(row[0], {})[row[1]] = float(row[2]) / float(row[3])
    out = {}
    for k, v in per.items():
        if len(v) == 3 and v["kernel_ns_under_pmc"] > 2e6:        # kernels above 2 ms
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0
            name = k.replace("void ", "").replace("natac::", "")
            out[name] = {"valu_wave_instructions": int(v["SQ_INSTS_VALU"]), "clock_ghz": round(cyc / v["kernel_ns_under_pmc"], 3),
                         "valu_issue_frac": round(v["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc), 3)}
    return out or None


# ------------------------------------------------------------------------------------------------ workloads
def make_workload(a, rank, world):
    """list of PackedChunks sub-batches of this rank + description"""
    from nucleoatac_amd.shard import balanced_ranges
    from nucleoatac_amd.synth import fragment_counts, make_synthetic_chunks
    if a.workload in ("cfg3", "cfg3-heavy", "cfg5"):
        nc = a.chunks or (12500 if a.workload == "cfg5" else 100000)
        L = a.chunk_len or 2120
        F = a.frags_per_chunk or 500
        if a.workload == "cfg5":
            pk = make_synthetic_chunks(nc, L, F, seed=a.seed + 1000 + rank)
            desc = ("configs[4]: one independent synthetic sample per rank (own seed), %d windows x 2 kb (L=%d), %d fragments, default VMat; "
                    "fp64 multinomial_cov tolerance sweep at the candidates of 200 chunks after the timed steps")
        elif a.workload == "cfg3":
            if world == 1:
                pk = make_synthetic_chunks(nc, L, F, seed=a.seed)
            else:       # N ranks share one host's cores: counter-seeded blocks on a few threads each (numpy releases the GIL)
                pk = _blocks_threaded(nc, L, F, [a.seed, 1000 + rank], max(1, min(4, _effective_cores() // world)))
            desc = "configs[2]: synthetic %d windows x 2 kb (L=%d after slop), %d fragments, default VMat 146x121, 1 GPU-shard per rank"
        else:
            counts = fragment_counts(nc, F, seed=a.seed + 1000 * rank, hot_frac=0.01, hot_mult=10)
            pk = make_synthetic_chunks(nc, L, F, seed=a.seed + 1000 * rank, counts=counts)
            desc = ("configs[2] windows with heavy-tailed load: %d windows (L=%d), Poisson(500) fragments per chunk, 1 %% of the "
                    "chunks 10x denser, %d fragments, default VMat 146x121, 1 GPU-shard per rank")
        return [pk], desc % (nc, L, pk.n_frags), dict(scaling="weak", imbalance=None, total_chunks=nc * world)
    # cfg4: strong scaling
    nc = a.chunks or 300000
    L = a.chunk_len or 10120
    F = a.frags_per_chunk or 667
    counts = fragment_counts(nc, F, seed=a.seed + 2)            # every rank: the same vector
    ranges = balanced_ranges(np.full(nc, L), np.concatenate(([0], np.cumsum(counts))), world)
    lo, hi = ranges[rank]
    BLOCK = 2000                                                # counter-seeded generation blocks
    subs = []
    for s0 in range(lo, hi, a.sub_chunks):
        s1 = min(hi, s0 + a.sub_chunks)
        parts = []
        for b0 in range((s0 // BLOCK) * BLOCK, s1, BLOCK):
            b1 = min(nc, b0 + BLOCK)
            blk = make_synthetic_chunks(b1 - b0, L, F, seed=[a.seed + 2, b0 // BLOCK], counts=counts[b0:b1], first_chunk=b0)
            x0, x1 = max(s0, b0) - b0, min(s1, b1) - b0
            parts.append(blk.subset(x0, x1) if (x0, x1) != (0, b1 - b0) else blk)
        subs.append(_concat(parts))
    per_rank = [(int((r1 - r0) * L), int(counts[r0:r1].sum())) for r0, r1 in ranges]
    bp = np.array([p[0] for p in per_rank], dtype=np.float64)
    fr = np.array([p[1] for p in per_rank], dtype=np.float64)
    imb = dict(bp_max_over_mean=round(float(bp.max() / bp.mean()), 4), fragments_max_over_mean=round(float(fr.max() / fr.mean()), 4),
               bp_per_rank=[int(x) for x in bp], fragments_per_rank=[int(x) for x in fr])
    desc = ("configs[3]: %d tiles x 10 kb (L=%d), %d fragments (Poisson(%d) per tile), chunk list sharded across %d rank(s) by "
            "sum(L) + 4 sum(F), sub-batches of <= %d chunks" % (nc, L, int(counts.sum()), F, world, a.sub_chunks))
    return subs, desc, dict(scaling="strong", imbalance=imb, total_chunks=nc)


def _blocks_threaded(nc, L, F, seed, n_threads, block=5000):
    """a configs[2]-shaped shard from counter-seeded blocks of `block` chunks, generated on n_threads threads"""
    from concurrent.futures import ThreadPoolExecutor
    from nucleoatac_amd.synth import make_synthetic_chunks
    starts = list(range(0, nc, block))
    gen = lambda b0: make_synthetic_chunks(min(block, nc - b0), L, F, seed=list(seed) + [b0 // block], first_chunk=b0)
    if n_threads <= 1:
        return _concat([gen(b0) for b0 in starts])
    with ThreadPoolExecutor(n_threads) as ex:
        return _concat(list(ex.map(gen, starts)))


def _concat(parts):
    from nucleoatac_amd.packing import PackedChunks
    if len(parts) == 1:
        return parts[0]
    offs = [np.zeros(1, np.int64)]
    boffs = [np.zeros(1, np.int64)]
    fo = bo = 0
    for p in parts:
        offs.append(p.frag_off[1:] + fo)
        boffs.append(p.bias_off[1:] + bo)
        fo += int(p.frag_off[-1])
        bo += int(p.bias_off[-1])
    return PackedChunks(chunk_start=np.concatenate([p.chunk_start for p in parts]),
                        chunk_len=np.concatenate([p.chunk_len for p in parts]), frag_off=np.concatenate(offs),
                        frag_lpos=np.concatenate([p.frag_lpos for p in parts]), frag_ilen=np.concatenate([p.frag_ilen for p in parts]),
                        bias_off=np.concatenate(boffs), bias_log=np.concatenate([p.bias_log for p in parts]))


def setup_ctx(device, par, sizes, nucp, nfrp):
    from nucleoatac_amd.device import Context
    ctx = Context(device)
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(sizes)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    return ctx


def bench_stages(tracks=()):
    """the stages of one step: nuc + occ + ins + candidate search (call_peaks, sep 25 / order 12 / boundary 60 as
    NucChunk.findAllNucs) with LR / var / z on the device; like the per-base tracks, the candidate arrays stay resident in
    HBM inside the timed region"""
    from nucleoatac_amd.executor import Stages
    return Stages(nuc_sd=10, occ=True, ins=(0, 2000), peaks=dict(min_signal=0, sep=25, boundary=60, order=12), tracks=tracks)


# ------------------------------------------------------------------------------------------------ host-to-host pipeline
def placement(pci):
    """where this rank's host side sits relative to its GPU: the CPUs the process may run on, their NUMA node(s), the GPU's NUMA node
    (sysfs).  The pinned slots of the host-to-host pipeline are allocated by these CPUs; on a two-socket node a rank whose CPUs are on
    the other socket than its GPU pays the inter-socket link on every transfer -- printed per rank so that a scaling run shows it."""
    def _cpulist(txt):
        out = set()
        for part in txt.strip().split(","):
            if part:
                lo, _, hi = part.partition("-")
                out.update(range(int(lo), int(hi or lo) + 1))
        return out
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else []
    nodes = {}
    try:
        for d in sorted(os.listdir("/sys/devices/system/node")):
            if d.startswith("node") and d[4:].isdigit():
                nodes[int(d[4:])] = _cpulist(open("/sys/devices/system/node/%s/cpulist" % d).read())
    except OSError:
        pass
    cpu_nodes = sorted(n for n, cs in nodes.items() if cs & set(cpus))
    gpu_node = None
    try:
        gpu_node = int(open("/sys/bus/pci/devices/%s/numa_node" % str(pci).lower()).read())
    except (OSError, ValueError):
        pass
    rng = "%d-%d" % (cpus[0], cpus[-1]) if cpus and cpus[-1] - cpus[0] + 1 == len(cpus) else ",".join(str(c) for c in cpus[:64])
    return dict(cpus_allowed=len(cpus), cpu_list=rng, cpu_numa_nodes=cpu_nodes, gpu_numa_node=gpu_node,
                gpu_local_to_cpus=None if gpu_node is None or gpu_node < 0 or not cpu_nodes else gpu_node in cpu_nodes)


def host_to_host(pk, device, par, sizes, nucp, nfrp, steps, sub_chunks, n_threads, as_text=False):
    """SURVEY.md section 8(d)'s boundary: packed inputs in host memory -> per-base tracks back in host memory, through the
    product's own executor (nucleoatac_amd/executor.py::PipelinedExecutor, the class `nucleoatac occ` / `nuc` run on): the
    chunk list is cut into sub-batches; `n_threads` host threads, each with its own natac context (stream) on the same GPU,
    take them in turn: upload (from pinned memory), all stages, download of the five default output tracks + the candidate
    arrays into pinned slots.  One untimed pass (allocations) precedes `steps` timed ones."""
    from nucleoatac_amd import _lib as L
    from nucleoatac_amd.device import pinned_copy
    from nucleoatac_amd.executor import PipelinedExecutor
    from nucleoatac_amd.packing import PackedChunks
    nsub = max(1, (pk.n_chunks + sub_chunks - 1) // sub_chunks)
    subs = []
    for i in range(nsub):
        s = pk.subset(i * sub_chunks, min(pk.n_chunks, (i + 1) * sub_chunks))
        subs.append(PackedChunks(chunk_start=s.chunk_start, chunk_len=s.chunk_len, frag_off=s.frag_off,
                                 frag_lpos=pinned_copy(s.frag_lpos), frag_ilen=pinned_copy(s.frag_ilen), bias_off=s.bias_off,
                                 bias_log=pinned_copy(s.bias_log), chroms=["chr%d" % (1 + (i * sub_chunks + k) % 22) for k in range(s.n_chunks)]))
    tracks = [getattr(L, t) for t in H2H_TRACKS]
    stages = bench_stages(tracks)
    if as_text:      # the same five tracks as finished bedGraph.gz bytes: Track.write_track + bgzip on the device
        from nucleoatac_amd.executor import Stages
        stages = Stages(nuc_sd=10, occ=True, ins=(0, 2000), peaks=dict(min_signal=0, sep=25, boundary=60, order=12), text_tracks=tracks)

    def configure(ctx):
        ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
        ctx.set_sizes(sizes)
        ctx.set_occ_model(nucp, nfrp, step=5, flank=60)

    with PipelinedExecutor(device, configure, stages, n_contexts=n_threads, slots_per_context=2) as ex:
        for r in ex.map((s, None) for s in subs):            # untimed pass: contexts, pool blocks, pinned slots
            r.release()
        ex.bytes_down = ex.bytes_up = 0
        t0 = time.perf_counter()
        ncand = 0
        for r in ex.map((s, None) for _ in range(steps) for s in subs):
            ncand += len(r.peaks[0])
            r.release()
        dt = time.perf_counter() - t0
        down, up = ex.bytes_down, ex.bytes_up
    return dict(host_to_host_mbp_s=round(pk.total_bp * steps / dt / 1e6, 2), seconds=round(dt, 3), steps=steps,
                sub_batches=nsub, chunks_per_sub_batch=sub_chunks, contexts=n_threads,
                executor="nucleoatac_amd.executor.PipelinedExecutor (product code)",
                tracks_downloaded=list(H2H_TRACKS) + ["candidates (chunk, pos, lr, var, z)"], candidates_per_step=ncand // max(1, steps),
                form="bedGraph.gz bytes (run-length text + BGZF formed on the device)" if as_text else "float64 arrays",
                gb_down_per_step=round(down / steps / 1e9, 3), gb_up_per_step=round(up / steps / 1e9, 3),
                pcie_gbs_down=round(down / dt / 1e9, 2), pcie_gbs_up=round(up / dt / 1e9, 2),
                note="pinned host buffers both ways; uploads, kernels and downloads of different sub-batches overlap")


# ------------------------------------------------------------------------------------------------ CLI end to end
def cli_end_to_end(n_chunks, cores, seed=0):
    """`nucleoatac run` (occ -> vprocess -> nuc -> merge -> nfr) as a user runs it, on a slice of the configs[2] workload written as input files
    (BED + reads .npz + genome .npz): files in -> .bedgraph.gz + .tbi + calls out, everything included (reading the inputs,
    PWM bias of the genome on the GPU, packing, the pipelined executor, the native writers, bgzip + tabix).  `nuc` runs with
    --cores for its per-nucleosome L-BFGS fits (host)."""
    import contextlib
    import shutil
    import tempfile
    from nucleoatac_amd.nucleoatac.cli import main as cli_main
    from nucleoatac_amd.synth import write_cli_dataset
    d = tempfile.mkdtemp(prefix="natac_cli_")
    try:
        bed, bam, fa = write_cli_dataset(d, n_chunks, 2120, 500, seed=seed)
        out = os.path.join(d, "e2e")
        bp = n_chunks * 2120
        with contextlib.redirect_stdout(sys.stderr):
            # `nucleoatac run` as a user runs it (one process: the occupancy tracks of step 1 stay resident in HBM for steps 3 and 5),
            # timed per step
            from nucleoatac_amd.nucleoatac import run_nfr as _rf, run_nuc as _rn, run_occ as _ro
            from nucleoatac_amd.nucleoatac.cli import nucleoatac_parser, run_chain
            secs, phases = {}, {}

            def on_step(name, seconds):
                secs[name] = seconds
                src = {"occ": _ro, "nuc": _rn, "nfr": _rf}.get(name)
                if src is not None:
                    phases[name] = dict(src.LAST_TIMINGS)

            run_chain(nucleoatac_parser().parse_args(["run", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--cores", str(cores)]),
                      on_step)
            t_occ, t_nuc, t_merge, t_nfr = secs["occ"], secs["nuc"], secs["merge"], secs["nfr"]
            occ_phases, nuc_phases, nfr_phases = phases["occ"], phases["nuc"], phases["nfr"]
            # `occ` once more from REAL input files: a coordinate-sorted .bam (both mates of every fragment) and a text .fa
            from nucleoatac_amd.synth import cli_dataset_as_real_files
            rbam, rfa = cli_dataset_as_real_files(bam, fa, d)
            t0 = time.perf_counter()
            cli_main(["occ", "--bed", bed, "--bam", rbam, "--fasta", rfa, "--out", out + "_real", "--cores", str(cores)])
            t_occ_real = time.perf_counter() - t0
            real = dict(occ_seconds=round(t_occ_real, 2), occ_mbp_s=round(bp / t_occ_real / 1e6, 2), bam_gb=round(os.path.getsize(rbam) / 1e9, 3),
                        read_bam_s=_ro.LAST_TIMINGS.get("read_bam"), note="the same windows from a real .bam (BGZF members inflated and records "
                        "walked on the GPU) and a text .fa (native loader) instead of the .npz stand-ins")
        size = lambda suffix: os.path.getsize(out + suffix)
        n_calls = sum(1 for _ in __import__("gzip").open(out + ".nucpos.bed.gz", "rt"))
        return dict(occ_mbp_s=round(bp / t_occ / 1e6, 2), nuc_mbp_s=round(bp / t_nuc / 1e6, 3), cores=cores, chunks=n_chunks, bp=bp,
                    occ_seconds=round(t_occ, 2), nuc_seconds=round(t_nuc, 2), merge_seconds=round(t_merge, 2), nfr_seconds=round(t_nfr, 2),
                    run_mbp_s=round(bp / sum(secs.values()) / 1e6, 3), run_seconds=round(sum(secs.values()), 2), nucleosome_calls=n_calls,
                    resident_occ_tracks="steps 3 and 5 read the occupancy tracks of step 1 from HBM, as the files show them (occstore.py); "
                                        "the files are written all the same",
                    occ_phases_s=occ_phases, nuc_phases_s=nuc_phases, nfr_phases_s=nfr_phases, real_inputs=real,
                    occ_track_bytes=sum(size("." + n + ".bedgraph.gz") for n in ("occ", "occ.lower_bound", "occ.upper_bound")),
                    nuc_track_bytes=sum(size("." + n + ".bedgraph.gz") for n in ("nucleoatac_signal", "nucleoatac_signal.smooth")),
                    out_dir=os.path.dirname(d) or "/tmp",
                    note="files in (BED, reads .npz, genome .npz) -> .bedgraph.gz + .tbi + .bed.gz out; occ writes 3 tracks + peaks, nuc "
                         "2 tracks + calls; nuc is bounded by the host L-BFGS fuzziness fits (one per call, --cores processes)")
    finally:
        shutil.rmtree(d, ignore_errors=True)


# ------------------------------------------------------------------------------------------------ main
def main():
    a = parse()
    what, world = resolve_ranks(a.gpus, os.environ)
    if what == "error":
        sys.stderr.write(world + "\n")
        sys.exit(2)
    if what == "spawn":
        sys.exit(spawn_ranks(world))
    a.gpus = world
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if "NATAC_DEVICE" in os.environ:          # the CLI's device override (nucleoatac_amd.get_context) holds for the bench too
        local_rank = int(os.environ["NATAC_DEVICE"])
    if a.share_device:
        local_rank = 0
        os.environ["NATAC_DEVICE"] = "0"
    if world > 1:
        # one shared helper for the CLI and the bench (nucleoatac_amd/shard.py): gloo control plane first, then an RCCL group
        # whose use is decided collectively; the only collectives are the contract's barrier and the max / sum of scalars
        from nucleoatac_amd import shard as dshard
        dist, _ = dshard.ensure_distributed(prefer=a.dist_backend, device=local_rank)
        a.dist_backend = dshard.control_backend()

    from nucleoatac_amd import _lib as L
    from nucleoatac_amd.synth import synth_occ_distributions, synth_size_distribution

    par = np.load(os.path.join(ROOT, "tests", "golden", "params_example.npz"))
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    t_gen = time.time()
    subs, desc, info = make_workload(a, rank, world)
    t_gen = time.time() - t_gen
    my_bp = sum(s.total_bp for s in subs)
    my_frags = sum(s.n_frags for s in subs)
    # CPU baseline first (rank 0, N=1 only): it forks worker processes, so run it before the HIP context exists
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline and subs:
        cpu = cpu_baseline(subs[0], par, sizes, nucp, nfrp, a.cpu_chunks, a.cpu_literal_chunks)

    from nucleoatac_amd.device import Context as _Ctx
    n_vis = _Ctx.device_count()
    if n_vis > 0 and local_rank >= n_vis:
        # a launcher that narrows every rank's view to its own GPU (ROCR_ / HIP_VISIBLE_DEVICES per rank) leaves ordinal 0 only: wrap, the
        # PCI bus ids gathered below still tell whether the ranks really sit on different GPUs
        local_rank %= n_vis
    ctx = setup_ctx(local_rank, par, sizes, nucp, nfrp)
    hip_dev, pci = ctx.device_ids()
    if dist is not None:
        # every rank must own a GPU of its own unless --share-device says otherwise: checked before any timing
        ids = [None] * world
        dist.all_gather_object(ids, (rank, local_rank, hip_dev, pci, os.environ.get("HIP_VISIBLE_DEVICES"), os.environ.get("ROCR_VISIBLE_DEVICES")))
        if not a.share_device and len({i[3] for i in ids}) < world:
            raise SystemExit("bench.py: ranks share a GPU (rank, local_rank, hip device, pci bus id, HIP_VISIBLE_DEVICES, "
                             "ROCR_VISIBLE_DEVICES): %s -- launch one rank per GPU or pass --share-device" % (ids,))
    from nucleoatac_amd.executor import ResidentShard
    t_up = time.time()
    # outputs of every sub-batch stay resident if they fit (~175 B per base incl. internal arrays); else they are recycled
    shard = ResidentShard(ctx, subs, recycle={"auto": "auto", "on": True, "off": False}[a.recycle])
    t_up = time.time() - t_up
    batches, recycle, last_n = shard.batches, shard.recycle, shard.last_n
    bg_flop, bg_tiles, bg_ext_tiles = bg_executed_flop(ctx, subs)
    stages = bench_stages()
    n_cand = [0]

    def step():
        n_cand[0] = shard.step(stages)

    on_gpu = dist is not None and a.dist_backend == "nccl"

    def barrier():
        ctx.sync()
        if dist is not None:
            dshard.barrier(sync_cuda=True)

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
    total_bp, total_frags, total_cand = my_bp, my_frags, n_cand[0]
    per_rank = None
    if dist is not None:
        my_dt = dt
        dt = dshard.all_reduce_scalars([dt], "max")[0]
        total_bp, total_frags, total_cand = (int(x) for x in dshard.all_reduce_scalars([my_bp, my_frags, n_cand[0]], "sum"))
        # per-rank timings so that a scaling run is diagnosable: who generated / uploaded / stepped how long
        rows = [None] * world
        dist.all_gather_object(rows, dict(rank=rank, generate_s=round(t_gen, 2), upload_s=round(t_up, 2),
                                          ms_per_step=round(my_dt / a.steps * 1e3, 3), bp=my_bp, fragments=my_frags,
                                          device=local_rank, hip_device=hip_dev, pci_bus_id=pci, placement=placement(pci)))
        per_rank = rows
    prof = ctx.profile()
    ms_per_step = dt / a.steps * 1e3
    value = total_bp * a.steps / dt / 1e6
    # one more (untimed) step under the shader-clock sampler: the clock every kernel class really ran at (fp64-dense kernels pull
    # the chip below its 2.4 GHz; peaks quoted at 2.4 GHz are not reachable by them)
    clocks = None
    if rank == 0 and not recycle:
        try:
            ctx.clock_trace_start(max_samples=20000, interval_us=50)
            step()
            ctx.sync()
            tr = ctx.clock_trace_stop()
            clocks = {k: round(v, 3) for k, v in tr["per_kernel"].items()}
            clocks["light_load_max"] = round(float(np.nanpercentile(tr["ghz"], 99)), 3) if np.isfinite(tr["ghz"]).any() else None
        except Exception as e:      # noqa: BLE001 -- a measurement aid must not cost the line
            clocks = {"error": "%s: %s" % (type(e).__name__, e)}

    # PCIe-inclusive figure of a resident batch (never `value`): one upload + one download of the per-base tracks that the
    # writers consume, pageable memory, no overlap -- the pipelined host-to-host rate follows below
    t_dn = None
    if not recycle and batches:
        t_dn = time.time()
        for t in (L.T_NORM, L.T_SMOOTH, L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER):
            batches[0].track(t)
        cand = batches[0].download_peaks(last_n[0])
        t_dn = time.time() - t_dn
        assert np.isfinite(cand[4][:1000]).any()
    cov_sweep = None
    if a.workload == "cfg5" and batches:
        # calculateCov variants (nucleoatac/multinomial_cov.pyx:20-31) at every candidate of 200 chunks of this rank's sample
        b0 = batches[0]
        cc, cp, _lr, var, _z = b0.run_peaks(min_signal=0, sep=25, boundary=60, order=12)
        ks = np.sort(np.random.default_rng(a.seed + rank).choice(subs[0].n_chunks, size=min(200, subs[0].n_chunks), replace=False))
        first, last = np.searchsorted(cc, ks, "left"), np.searchsorted(cc, ks, "right")
        sel = np.concatenate([np.arange(x, y) for x, y in zip(first, last)]) if len(ks) else np.zeros(0, np.int64)
        t0 = time.perf_counter()
        lit = b0.run_candidates_cov(cc[sel], cp[sel], "literal")
        t_lit = time.perf_counter() - t0
        t0 = time.perf_counter()
        clo = b0.run_candidates_cov(cc[sel], cp[sel], "closed")
        t_clo = time.perf_counter() - t0
        f32 = b0.run_candidates_cov(cc[sel], cp[sel], "fp32")
        ok = lit > 0
        rel = lambda x: float(np.max(np.abs(x[ok] - lit[ok]) / lit[ok])) if ok.any() else 0.0
        mine = dict(rank=rank, candidates=int(len(sel)), closed_fp64=rel(clo), fp32=rel(f32), run_peaks_var=rel(var[sel]),
                    literal_s=round(t_lit, 3), closed_s=round(t_clo, 3))
        rows = [mine]
        if dist is not None:
            rows = [None] * world
            dist.all_gather_object(rows, mine)
        cov_sweep = dict(samples=rows, worst_closed_fp64=max(r["closed_fp64"] for r in rows), worst_fp32=max(r["fp32"] for r in rows),
                         reference="the device's literal O(N^2) fp64 pair sum (the .pyx's own terms); tests pin it to the oracle's C restatement")
    shard.close()
    h2h = None
    if world > 1 and a.h2h_ranks and a.workload != "cfg4":
        # first contact of the host-to-host pipeline with SEVERAL ranks on one host (pinned slots, six contexts per rank, the ranks'
        # executors next to each other): every rank runs the float64 form on its own shard at the same time; the rates go into per_rank.
        # Not part of the driver's N > 1 line by default (--h2h-ranks): the contract's value is the HBM-resident step above.
        ctx.close()
        ctx = None
        dshard.barrier(sync_cuda=False)
        mine = host_to_host(subs[0].subset(0, min(subs[0].n_chunks, a.h2h_rank_chunks)), local_rank, par, sizes, nucp, nfrp,
                            max(1, min(a.steps, 3)), a.h2h_sub, a.h2h_threads)
        rows = [None] * world
        dist.all_gather_object(rows, (rank, mine["host_to_host_mbp_s"], mine["pcie_gbs_down"], mine["seconds"]))
        for r, v, g, sec in rows:
            per_rank[r].update(host_to_host_mbp_s=v, host_to_host_pcie_gbs_down=g, host_to_host_seconds=sec)
    if rank == 0 and world == 1 and not a.no_h2h and a.workload != "cfg4":
        ctx.close()
        ctx = None
        h2h = host_to_host(subs[0], local_rank, par, sizes, nucp, nfrp, a.steps, a.h2h_sub, a.h2h_threads)
        # the two forms are bound by different things (float64: the PCIe link; bedGraph.gz: ~25 small launches and a few synchronisations per
        # track, which more contexts with larger sub-batches hide better -- profiles/r6/README.md, the sweeps), so each runs with the executor
        # parameters that suit it; both are printed in the form's own `contexts` / `chunks_per_sub_batch`.  An explicit --h2h-sub /
        # --h2h-threads applies to both forms.
        explicit = any(x.startswith(("--h2h-sub", "--h2h-threads")) for x in sys.argv[1:])
        t_sub = a.h2h_text_sub or (a.h2h_sub if explicit else 5000)
        t_thr = a.h2h_text_threads or (a.h2h_threads if explicit else 10)
        h2h["as_bedgraph_gz"] = host_to_host(subs[0], local_rank, par, sizes, nucp, nfrp, a.steps, t_sub, t_thr, as_text=True)
    e2e = None
    if rank == 0 and world == 1 and a.cli_chunks > 0 and a.workload == "cfg3":
        if ctx is not None:
            ctx.close()
            ctx = None
        e2e = cli_end_to_end(a.cli_chunks, _effective_cores(), seed=a.seed)

    if rank == 0:
        # roofline of the dominant kernel class of this run (largest HIP-event time on the launch stream)
        dom = max(("background", "occ_mle", "candidates"), key=lambda k: prof[k][0])
        dom_ms, dom_n = prof[dom]
        dom_avg_s = (dom_ms / max(1, dom_n)) / 1e3
        launches_per_step = max(1, len(batches))
        bp_per_launch = my_bp / launches_per_step
        traffic = pmc_traffic_bytes(KERNEL_SYMBOL[dom]) if a.workload == "cfg3" else None
        alg_bytes = ALG_BYTES_PER_BP[a.workload] * bp_per_launch
        achieved = alg_bytes / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
        bg_ms, bg_n = prof["background"]
        bg_avg_s = (bg_ms / max(1, bg_n)) / 1e3
        direct_tflops = FLOP_PER_BP_BG * bp_per_launch / bg_avg_s / 1e12 if bg_avg_s > 0 else 0.0
        fft_tflops = bg_flop / launches_per_step / bg_avg_s / 1e12 if bg_avg_s > 0 else 0.0
        step_gbs = ALG_BYTES_PER_BP[a.workload] * my_bp / (dt / a.steps) / 1e9
        valu = pmc_valu_issue() if a.workload == "cfg3" else None
        dom_issue = None
        if valu:
            hit = [v["valu_issue_frac"] for k, v in valu.items() if KERNEL_SYMBOL[dom] in k]
            dom_issue = max(hit) if hit else None
        st = pmc_step_traffic() if a.workload == "cfg3" else None
        step_alg = ALG_BYTES_PER_BP[a.workload] * my_bp
        step_traffic = None
        if st:
            raw, x2 = st["fetch_bytes"] + st["write_bytes"], 2 * st["fetch_bytes"] + st["write_bytes"]
            step_traffic = {"fetch_plus_write_bytes": raw, "with_fetch_x2_bytes": x2, "algorithmic_bytes": step_alg,
                            "ratio_raw": round(raw / step_alg, 3), "ratio_fetch_x2": round(x2 / step_alg, 3),
                            "note": "sum over ALL kernels of one step (committed rocprofv3 --pmc passes, see traffic_source); "
                                    "algorithmic = %.1f B/bp x the step's bases" % ALG_BYTES_PER_BP[a.workload]}
        fp64_bound = dom == "background"
        out = {
            "metric": "Mbp/s through occ+nuc signal pipeline", "value": round(value, 3), "unit": "Mbp/s",
            "value_boundary": "hbm_resident: packed inputs and every output track stay in HBM inside the timed region; SURVEY.md 8(d)'s "
                              "boundary (packed inputs in host memory -> output tracks back in host memory, PCIe-bound) is value_host_to_host",
            "value_host_to_host": None if h2h is None else h2h["host_to_host_mbp_s"],
            "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": info["scaling"], "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": desc, "chunks_total": info["total_chunks"], "chunks_this_rank": int(sum(s.n_chunks for s in subs)),
                       "bp_total": total_bp, "fragments_total": total_frags, "candidates_per_step": total_cand,
                       "sub_batches_this_rank": len(subs), "outputs_recycled": bool(recycle),
                       "sharding": "chunk list split across ranks, no collective", "shard_imbalance": info["imbalance"]},
            # the dominant kernel is bound by fp64 vector issue (+ LDS transposes), not by HBM (~5e4 flop per base against
            # ~80 bytes): achieved / peak are its executed fp64 rate against the fp64 vector peak; the HBM view the metric's name asks
            # for is kept under "hbm" (same launch time, algorithmic bytes of the whole pipeline per launch)
            "roofline": {"bound": "fp64_valu" if fp64_bound else "hbm", "kernel": KERNEL_LABEL[dom],
                         "achieved": round(fft_tflops, 2) if fp64_bound else round(achieved, 2),
                         "peak": FP64_PEAK_TFLOPS if fp64_bound else HBM_PEAK_GBS, "unit": "TFLOP/s" if fp64_bound else "GB/s",
                         "frac": round(fft_tflops / FP64_PEAK_TFLOPS, 5) if fp64_bound else round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": traffic,
                         "traffic_note": "HBM bytes of THIS kernel per launch (FETCH_SIZE + WRITE_SIZE, committed PMC pass); the whole "
                                         "step's bytes against the algorithmic bytes are under step_traffic",
                         "hbm": {"achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(achieved / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": alg_bytes},
                         "step_traffic": step_traffic,
                         "clock_ghz": clocks,
                         "traffic_source": pmc_source() if a.workload == "cfg3" else None,
                         # the meaningful fraction for this kernel: share of the SIMDs' issue cycles its fp64 VALU stream uses
                         "fp64_issue_frac": dom_issue,
                         "avg_launch_ms": round(dom_avg_s * 1e3, 3), "launches": int(dom_n),
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "whole_step": {"algorithmic_gb": round(ALG_BYTES_PER_BP[a.workload] * my_bp / 1e9, 3),
                                        "achieved_gbs": round(step_gbs, 2), "frac": round(step_gbs / HBM_PEAK_GBS, 5)},
                         "note": "the path is fp64-VALU / LDS bound, not HBM bound (SURVEY 8d: ~5e4 flop/base against ~80 B/base); "
                                 "background_fp64 gives the arithmetic rate of the background kernel",
                         "background_fp64": {"avg_launch_ms": round(bg_avg_s * 1e3, 3),
                                             "direct_equivalent_tflops": round(direct_tflops, 2),
                                             "executed_tflops": round(fft_tflops, 2), "peak": FP64_PEAK_TFLOPS,
                                             "executed_flop_per_bp": round(bg_flop / max(1, my_bp), 1),
                                             "tiles": int(bg_tiles), "extended_tiles": int(bg_ext_tiles),
                                             "kernels": "natac_background_fft (transforms + the edge pass of its extended tiles)",
                                             "unit": "TFLOP/s", "frac_executed": round(fft_tflops / FP64_PEAK_TFLOPS, 4)}},
            "kernels_ms_per_step": {k: round(v[0] / a.steps, 3) for k, v in prof.items()},
            "per_rank": per_rank, "placement": placement(pci), "control_plane": a.dist_backend if dist is not None else None,
            "valu_issue_from_committed_pmc": valu,
            "host": {"generate_s": round(t_gen, 2), "upload_s": round(t_up, 2),
                     "download_5_tracks_and_candidates_s": None if t_dn is None else round(t_dn, 2),
                     "pcie_inclusive_mbp_s_no_overlap": None if t_dn is None else round(
                         subs[0].total_bp / (dt / a.steps / max(1, len(subs)) + t_up / max(1, len(subs)) + t_dn) / 1e6, 2)},
        }
        if h2h is not None:
            out["host_to_host"] = h2h
        if e2e is not None:
            out["cli_end_to_end"] = e2e
        if cov_sweep is not None:
            out["multinomial_cov_sweep"] = cov_sweep
        if cpu is not None:
            cpu["reference_calibration"] = reference_calibration()
            out["cpu_baseline"] = cpu
        assert out["n_gpus"] == world == a.gpus and (per_rank is None or sorted(r["rank"] for r in per_rank) == list(range(world)))
        print(json.dumps(out))
        sys.stdout.flush()
    if ctx is not None:
        ctx.close()
    if dist is not None:
        dshard.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
