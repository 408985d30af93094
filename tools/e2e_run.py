#!/usr/bin/env python3
"""`nucleoatac run` (occ -> vprocess -> nuc -> merge -> nfr) end to end at any size, one GPU, with the per-phase seconds of every
driver: N windows of length L with F fragments each written as input files -- with `real`: a coordinate-sorted .bam (two 50-base
reads per fragment) and a text .fa, else the .npz stand-ins -- then the five steps as the command line runs them.

  python tools/e2e_run.py 60000 10120 667 /dev/shm real     # one GPU's share of BASELINE configs[3] (300 k x 10 kb tiles over 8 GPUs)

Prints one JSON line (committed under profiles/)."""
import contextlib
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10120
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 667
    base = sys.argv[4] if len(sys.argv) > 4 else None
    real = len(sys.argv) > 5 and sys.argv[5] == "real"
    import bench
    from nucleoatac_amd import occstore
    from nucleoatac_amd.nucleoatac import run_nfr as rf, run_nuc as rn, run_occ as ro
    from nucleoatac_amd.nucleoatac.cli import main as cli_main
    from nucleoatac_amd.synth import cli_dataset_as_real_files, write_cli_dataset
    cores = bench._effective_cores()
    d = tempfile.mkdtemp(prefix="natac_e2e_run_", dir=base)
    try:
        t0 = time.perf_counter()
        bed, bam, fa = write_cli_dataset(d, n, L, F, seed=0)
        if real:
            bam, fa = cli_dataset_as_real_files(bam, fa, d)
        t_gen = time.perf_counter() - t0
        out = os.path.join(d, "e2e")
        bp = n * L
        sec, phases = {}, {}

        from nucleoatac_amd.nucleoatac.cli import nucleoatac_parser, run_chain
        resident = {}

        def on_step(name, seconds):
            sec[name] = round(seconds, 2)
            src = {"occ": ro, "nuc": rn, "nfr": rf}.get(name)
            if src is not None:
                phases[name] = dict(src.LAST_TIMINGS)
            st = occstore.lookup(out + ".occ.bedgraph.gz")
            if st is not None:
                resident.update(st.dev.info(), regions_served=st.reads)

        with contextlib.redirect_stdout(sys.stderr):
            t_all = time.perf_counter()
            run_chain(nucleoatac_parser().parse_args(["run", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--cores", str(cores)]),
                      on_step)
            total = time.perf_counter() - t_all
        resident = resident or None
        size = lambda s: os.path.getsize(out + s) if os.path.exists(out + s) else None
        import gzip
        n_calls = sum(1 for _ in gzip.open(out + ".nucpos.bed.gz", "rt"))
        print(json.dumps(dict(
            chunks=n, chunk_len=L, fragments_per_chunk=F, bp=bp, cores=cores, run_seconds=round(total, 2), run_mbp_s=round(bp / total / 1e6, 3),
            step_seconds=sec, step_mbp_s={k: round(bp / v / 1e6, 2) for k, v in sec.items() if v > 0}, phases_s=phases,
            nucleosome_calls=n_calls, resident_occ_tracks=resident,
            bytes=dict(bam=os.path.getsize(bam), fasta=os.path.getsize(fa),
                       occ_tracks=sum(size("." + x + ".bedgraph.gz") for x in ("occ", "occ.lower_bound", "occ.upper_bound")),
                       nuc_tracks=sum(size("." + x + ".bedgraph.gz") for x in ("nucleoatac_signal", "nucleoatac_signal.smooth")),
                       ins_track=size(".ins.bedgraph.gz"), nucpos=size(".nucpos.bed.gz"), nfrpos=size(".nfrpos.bed.gz")),
            generate_inputs_s=round(t_gen, 1), inputs="real .bam + text .fa" if real else ".npz stand-ins", out_dir=d)))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
