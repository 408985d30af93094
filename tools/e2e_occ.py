"""`nucleoatac occ` end to end on N chunks of the configs[2] workload written as input files (bench.py's cli_end_to_end runs the
10,000-chunk slice; this runs any size, occ only):  python tools/e2e_occ.py 100000 [/dev/shm] [real]
`real`: the inputs are a real coordinate-sorted .bam (two 50-base reads per fragment) and a text .fa instead of the .npz stand-ins."""
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
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    base = sys.argv[2] if len(sys.argv) > 2 else None
    from nucleoatac_amd.nucleoatac import run_occ as ro
    from nucleoatac_amd.nucleoatac.cli import main as cli_main
    from nucleoatac_amd.synth import write_cli_dataset
    d = tempfile.mkdtemp(prefix="natac_e2e_", dir=base)
    try:
        t0 = time.perf_counter()
        bed, bam, fa = write_cli_dataset(d, n, 2120, 500, seed=0)
        real = len(sys.argv) > 3 and sys.argv[3] == "real"
        if real:
            from nucleoatac_amd.synth import cli_dataset_as_real_files
            bam, fa = cli_dataset_as_real_files(bam, fa, d)
        t_gen = time.perf_counter() - t0
        out = os.path.join(d, "e2e")
        with contextlib.redirect_stdout(sys.stderr):
            t0 = time.perf_counter()
            cli_main(["occ", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out])
            dt = time.perf_counter() - t0
        size = sum(os.path.getsize(out + "." + x + ".bedgraph.gz") for x in ("occ", "occ.lower_bound", "occ.upper_bound"))
        print(json.dumps(dict(chunks=n, bp=n * 2120, occ_seconds=round(dt, 2), occ_mbp_s=round(n * 2120 / dt / 1e6, 2), phases_s=dict(ro.LAST_TIMINGS),
                              track_bytes=size, write_gb_s=round(size / dt / 1e9, 2), out_dir=d, generate_inputs_s=round(t_gen, 1),
                              device_writer=ro.DEVICE_WRITER, inputs="real .bam (%.2f GB) + .fa" % (os.path.getsize(bam) / 1e9) if real else ".npz stand-ins")))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
