"""cProfile of `nucleoatac nfr` after occ / vprocess / nuc / merge on N chunks of the configs[2] workload (GPU box):
python tools/profile_nfr.py 10000"""
import contextlib
import cProfile
import os
import pstats
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    from nucleoatac_amd.nucleoatac.cli import main as cli_main
    from nucleoatac_amd.synth import write_cli_dataset
    d = tempfile.mkdtemp(prefix="natac_nfr_")
    try:
        bed, bam, fa = write_cli_dataset(d, n, 2120, 500, seed=0)
        out = os.path.join(d, "e2e")
        with contextlib.redirect_stdout(sys.stderr):
            cli_main(["occ", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--cores", "16"])
            cli_main(["vprocess", "--sizes", out + ".nuc_dist.txt", "--out", out])
            cli_main(["nuc", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--cores", "16", "--occ_track",
                      out + ".occ.bedgraph.gz", "--vmat", out + ".VMat", "--sizes", out + ".fragmentsizes.txt"])
            cli_main(["merge", "--occpeaks", out + ".occpeaks.bed.gz", "--nucpos", out + ".nucpos.bed.gz", "--out", out])
            pr = cProfile.Profile()
            pr.enable()
            cli_main(["nfr", "--bed", bed, "--occ_track", out + ".occ.bedgraph.gz", "--calls", out + ".nucmap_combined.bed.gz",
                      "--out", out, "--fasta", fa, "--bam", bam])
            pr.disable()
        pstats.Stats(pr).sort_stats("tottime").print_stats(25)
        from nucleoatac_amd.nucleoatac import run_nfr
        print(run_nfr.LAST_TIMINGS)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
