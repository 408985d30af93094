"""Host-side profile of the `occ` / `nuc` drivers on a synthetic genome (development tool).
usage: python tools/profile_cli.py [chrom_len] [n_chunks] [cores]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import golden, synth_genome  # noqa: E402


def main():
    chrom_len = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
    n_chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    cores = sys.argv[3] if len(sys.argv) > 3 else "1"
    from nucleoatac_amd.nucleoatac.cli import main as cli
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    from nucleoatac_amd.pyatac.fragmentsizes import FragmentSizes
    par = golden("params_example")
    l, n, seq = synth_genome(5, chrom_len=chrom_len)
    td = tempfile.mkdtemp()
    bam = os.path.join(td, "s.npz")
    FragmentStore(["chrS"], [len(seq)], {"chrS": l - 4}, {"chrS": n + 8}).save_npz(bam)
    fa = os.path.join(td, "s.fa")
    with open(fa, "w") as f:
        f.write(">chrS\n" + seq.tobytes().decode() + "\n")
    bed = os.path.join(td, "r.bed")
    step = (chrom_len - 4000) // n_chunks
    with open(bed, "w") as f:
        for i in range(n_chunks):
            s = 2000 + i * step
            f.write("chrS\t%d\t%d\n" % (s, s + min(2000, step - 130)))
    sizes = os.path.join(td, "sizes.txt")
    FragmentSizes(0, 251, vals=par["sizes"]).save(sizes)
    vm = os.path.join(td, "v.npz")
    np.savez(vm, vmat=par["vmat"], vlower=par["vlower"], vupper=par["vupper"])
    out = os.path.join(td, "o")
    common = ["--bed", bed, "--bam", bam, "--fasta", fa, "--sizes", sizes, "--out", out]
    for sub in (["occ"] + common, ["nuc"] + common + ["--vmat", vm, "--occ_track", out + ".occ.bedgraph.gz", "--cores", cores]):
        pr = cProfile.Profile()
        t = time.time()
        pr.enable()
        cli(sub)
        pr.disable()
        print("=== %s: %.1f s for %d chunks" % (sub[0], time.time() - t, n_chunks))
        pstats.Stats(pr).sort_stats("cumulative").print_stats(22)


if __name__ == "__main__":
    main()
