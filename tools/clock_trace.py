#!/usr/bin/env python3
"""Shader clock during the configs[2] step: the one-wave sampler of natac_clock_trace_* runs next to `steps` steps of the bench
workload and the profiled launches are laid over its time axis.  Prints the mean clock per kernel class and a 1-ms-binned series
of the last step as one JSON line.

usage (GPU box):  python tools/clock_trace.py [n_chunks] [steps] > gpurun_out/clock_trace.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    from nucleoatac_amd.executor import ResidentShard
    from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    par = np.load(os.path.join(ROOT, "tests", "golden", "params_example.npz"))
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    pk = make_synthetic_chunks(nc, 2120, 500, seed=0)
    ctx = bench.setup_ctx(0, par, sizes, nucp, nfrp)
    shard = ResidentShard(ctx, [pk], recycle=False)
    stages = bench.bench_stages()
    for _ in range(2):
        shard.step(stages)
    ctx.sync()
    ctx.profile_enable(True)
    ctx.profile_reset()
    ctx.clock_trace_start(max_samples=40000, interval_us=50)
    for _ in range(steps):
        shard.step(stages)
    ctx.sync()
    tr = ctx.clock_trace_stop()
    prof = ctx.profile()
    iv = tr["intervals"]
    # the last step = from its frag_gather launch to the end of the trace
    g = [a for k, a, b in iv if k == "frag_gather"]
    t_last = g[-1] if g else 0.0
    mid = 0.5 * (tr["t_ms"][1:] + tr["t_ms"][:-1])
    m = mid >= t_last
    bins = np.floor(mid[m] - t_last).astype(int)
    series = [round(float(np.nanmean(tr["ghz"][m][bins == b])), 3) for b in range(int(bins.max()) + 1)] if m.any() else []
    out = dict(chunks=nc, steps=steps, samples=int(len(tr["t_ms"])),
               clock_ghz_per_kernel={k: round(v, 3) for k, v in tr["per_kernel"].items()},
               kernels_ms_per_step={k: round(v[0] / steps, 3) for k, v in prof.items()},
               last_step_intervals=[(k, round(a - t_last, 3), round(b - t_last, 3)) for k, a, b in iv if a >= t_last - 1e-6],
               last_step_ghz_per_ms=series,
               idle_clock_ghz=round(float(np.nanpercentile(tr["ghz"], 99)), 3) if np.isfinite(tr["ghz"]).any() else None)
    shard.close()
    ctx.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
