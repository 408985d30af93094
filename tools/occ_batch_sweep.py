import contextlib, json, os, sys, time, tempfile, shutil
sys.path.insert(0, os.getcwd())
from nucleoatac_amd.nucleoatac import run_occ as ro
from nucleoatac_amd.nucleoatac.cli import main as cli_main
from nucleoatac_amd.synth import write_cli_dataset, cli_dataset_as_real_files
n, L, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = tempfile.mkdtemp(prefix="natac_occ_only_", dir="/dev/shm")
try:
    bed, bam, fa = write_cli_dataset(d, n, L, F, seed=0)
    for bc in sys.argv[4:]:
        ro.BATCH_CHUNKS = int(bc)
        out = os.path.join(d, "o%s" % bc)
        with contextlib.redirect_stdout(sys.stderr):
            t0 = time.perf_counter()
            cli_main(["occ", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out])
            dt = time.perf_counter() - t0
        print(json.dumps(dict(batch_chunks=int(bc), occ_seconds=round(dt, 2), mbp_s=round(n * L / dt / 1e6, 1), phases=dict(ro.LAST_TIMINGS))))
        for f in os.listdir(d):
            if f.startswith("o%s." % bc):
                os.remove(os.path.join(d, f))
finally:
    shutil.rmtree(d, ignore_errors=True)
