#!/usr/bin/env python3
"""Calibration of bench.py's `cpu_baseline` against THE REFERENCE ITSELF (BASELINE.md "Calibration in this container").

bench.py times the oracle (a port) in the reference's Pool.map shape.  This script runs, on IDENTICAL inputs and one core,
  * the reference's own OccChunk.process + NucChunk.process (Occupancy.py:241-247, NucleosomeCalling.py:328-340; scratch py3 copy
    made by oracle/make_scratch_ref.py, reads / genome from the .npz pysam stand-in), and
  * bench._cpu_chunk in its "literal" and "optimised" modes,
on chunks of the configs[2] shape (2,120 bp, ~500 fragments each), and writes seconds per chunk + the ratios to
profiles/r4/cpu_baseline_calibration.json, which bench.py quotes as `cpu_baseline.reference_calibration`.
Runs only in the build container (needs /root/reference).

usage:  python oracle/make_scratch_ref.py /tmp/natac_scratch_ref
        python tests/golden/calibrate_cpu_baseline.py [/tmp/natac_scratch_ref] [n_chunks]
"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
SCRATCH = sys.argv[1] if len(sys.argv) > 1 else "/tmp/natac_scratch_ref"
N_CHUNKS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.argv = [sys.argv[0], SCRATCH]
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402  (sets up the scratch reference's import path)

spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(REPO, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    vmat, fd, pwm = MG.case_params()
    L = 2120
    starts = [1500 + 2600 * k for k in range(N_CHUNKS)]
    chrom_len = starts[-1] + L + 2000
    # configs[2] density: ~500 fragments per 2,120-bp chunk (make_synth_genome draws 0.35 per base: thin to 500 / 2,372)
    bam, fa, l_all, n_all, seq = MG.make_synth_genome(77, chrom_len=chrom_len)
    rng = np.random.default_rng(78)
    keep = rng.random(len(l_all)) < (500.0 / (L + 252)) / 0.35
    l_all, n_all = l_all[keep], n_all[keep]
    np.savez(bam, chrom_names=np.array(["chrS"]), chrom_lengths=np.array([chrom_len]), pos_chrS=l_all - 4, tlen_chrS=n_all + 8)
    sizes = MG.FragmentSizes(0, 251, vals=fd.fragmentsizes.get(0, 251))
    oparams = MG.Occ.OccupancyParameters(fd, 251, fa, "Human", sep=120, min_occ=0.1, flank=60, bam=bam, ci=0.9, step=5)
    nparams = MG.Nuc.NucParameters(vmat=vmat, fragmentsizes=sizes, bam=bam, fasta=fa, pwm="Human", occ_track=None, sd=10,
                                   nonredundant_sep=120, redundant_sep=25, min_z=3, min_lr=0, atac=True)
    t_occ = t_nuc = 0.0
    tasks = []
    n_frags = []
    for s in starts:
        e = s + L
        ch = MG.Chunk("chrS", s, e)
        t0 = time.perf_counter()
        occ = MG.Occ.OccChunk(ch)
        occ.process(oparams)
        t_occ += time.perf_counter() - t0
        t0 = time.perf_counter()
        nuc = MG.Nuc.NucChunk(MG.Chunk("chrS", s, e))
        nuc.process(nparams)
        t_nuc += time.perf_counter() - t0
        bt = MG.InsertionBiasTrack("chrS", s - 246, e + 247, log=True)
        bt.computeBias(fa, oparams.chrs, pwm)
        l, n = MG.frags_for_chunk(l_all, n_all, s, e)
        n_frags.append(int(((l + (n - 1) // 2 >= s - 126) & (l + (n - 1) // 2 < e + 126)).sum()))
        tasks.append([(l - s).astype(np.int64), n.astype(np.int64), L, bt.vals.copy(), 246, vmat.mat, int(vmat.lower), int(vmat.upper),
                      sizes.get(0, 251), oparams.occ_calc_params.nuc_probs, oparams.occ_calc_params.nfr_probs])
    port = {}
    for mode, literal in (("literal", True), ("optimised", False)):
        bench._cpu_chunk(tuple(tasks[0] + [literal]))                # warm-up (imports, scipy plans)
        t0 = time.perf_counter()
        for t in tasks:
            bench._cpu_chunk(tuple(t + [literal]))
        port[mode] = (time.perf_counter() - t0) / len(tasks)
    ref = (t_occ + t_nuc) / len(tasks)
    out = dict(chunks=len(tasks), chunk_len=L, fragments_per_chunk=n_frags, cores=1,
               reference_s_per_chunk=round(ref, 3), reference_occ_s_per_chunk=round(t_occ / len(tasks), 3),
               reference_nuc_s_per_chunk=round(t_nuc / len(tasks), 3),
               port_literal_s_per_chunk=round(port["literal"], 3), port_optimised_s_per_chunk=round(port["optimised"], 3),
               reference_over_port_literal=round(ref / port["literal"], 2), reference_over_port_optimised=round(ref / port["optimised"], 2),
               note="same chunks, one core, build container; the port is FASTER than the reference it stands in for (it skips the unused "
                    "getIns correlate2d, getFuzz and makeInsertionTrack of NucChunk.process), so bench.py's cpu_baseline is a conservative "
                    "(too fast) stand-in for the reference's CPU path",
               reference="OccChunk.process + NucChunk.process of /root/reference (py3 scratch copy, oracle/make_scratch_ref.py)",
               generated_by="tests/golden/calibrate_cpu_baseline.py")
    dst = os.path.join(REPO, "profiles", "r4", "cpu_baseline_calibration.json")
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
