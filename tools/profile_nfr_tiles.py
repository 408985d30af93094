"""cProfile of `nucleoatac nfr` inside `nucleoatac run` on N tiles of 10,120 bp (GPU box):  python tools/profile_nfr_tiles.py 20000"""
import contextlib, cProfile, os, pstats, shutil, sys, tempfile
sys.path.insert(0, os.getcwd())
from nucleoatac_amd.nucleoatac.cli import nucleoatac_parser, run_chain
from nucleoatac_amd.nucleoatac import run_nfr as rf
from nucleoatac_amd.synth import write_cli_dataset


def main():
    n = int(sys.argv[1])

    d = tempfile.mkdtemp(prefix="natac_nfrp_", dir="/dev/shm")
    try:
        bed, bam, fa = write_cli_dataset(d, n, 10120, 667, seed=0)
        out = os.path.join(d, "e2e")
        args = nucleoatac_parser().parse_args(["run", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--cores", "16"])
        pr = cProfile.Profile()
        real = rf.run_nfr
        def wrapped(a):
            pr.enable()
            try:
                return real(a)
            finally:
                pr.disable()
        import nucleoatac_amd.nucleoatac.run_nfr as m
        m.run_nfr = wrapped
        with contextlib.redirect_stdout(sys.stderr):
            run_chain(args)
        pstats.Stats(pr).sort_stats("tottime").print_stats(22)
        print(rf.LAST_TIMINGS)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":      # the fit pool of `nuc` spawns workers that import this file
    main()
