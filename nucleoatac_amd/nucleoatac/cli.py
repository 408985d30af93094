"""`nucleoatac run | occ | vprocess | nuc | merge | nfr` command line with the reference's flag names and defaults
(nucleoatac/cli.py:7-330)."""
import argparse
import sys


def add_occ_parser(sub):
    p = sub.add_parser("occ", help="Call nucleosome occupancy")
    p.add_argument("--bed", required=True, help="Peaks in bed format")
    p.add_argument("--bam", required=True, help="Sorted BAM file (or a FragmentStore .npz)")
    p.add_argument("--out", required=True, help="output basename")
    p.add_argument("--fasta", help="genome fasta (if absent, bias is not calculated)")
    p.add_argument("--pwm", default="Human", help="Tn5 PWM name or descriptor file")
    p.add_argument("--sizes", help="file with fragment size distribution")
    p.add_argument("--cores", type=int, default=1, help="accepted for compatibility; GPU batches replace the pool")
    p.add_argument("--upper", type=int, default=251)
    p.add_argument("--flank", type=int, default=60)
    p.add_argument("--min_occ", type=float, default=0.1)
    p.add_argument("--nuc_sep", type=int, default=120)
    p.add_argument("--confidence_interval", type=float, default=0.9)
    p.add_argument("--step", type=int, default=5)


def add_nuc_parser(sub):
    p = sub.add_parser("nuc", help="Call nucleosome positions and make signal tracks")
    p.add_argument("--bed", required=True)
    p.add_argument("--vmat", required=True, help="VMat file (text descriptor or .npz)")
    p.add_argument("--bam", required=True)
    p.add_argument("--out", required=True)
    p.add_argument("--fasta")
    p.add_argument("--pwm", default="Human")
    p.add_argument("--sizes")
    p.add_argument("--occ_track")
    p.add_argument("--cores", type=int, default=1, help="host processes for the per-nucleosome fuzziness fits")
    p.add_argument("--write_all", action="store_true", default=False)
    p.add_argument("--not_atac", dest="atac", action="store_false", default=True)
    p.add_argument("--min_z", type=float, default=3)
    p.add_argument("--min_lr", type=float, default=0)
    p.add_argument("--nuc_sep", type=int, default=120)
    p.add_argument("--redundant_sep", type=int, default=25)
    p.add_argument("--sd", type=int, default=10)


def add_vprocess_parser(sub):
    p = sub.add_parser("vprocess", help="Make processed vplot to use for nucleosome calling")
    p.add_argument("--out", required=True)
    p.add_argument("--sizes", help="Insert distribution file")
    p.add_argument("--vplot", default=None, help="VMat file. Default is the Vplot from S. cer. shipped with the package")
    p.add_argument("--lower", type=int, default=105)
    p.add_argument("--upper", type=int, default=251)
    p.add_argument("--flank", type=int, default=60)
    p.add_argument("--smooth", type=float, default=0.75)
    p.add_argument("--plot_extra", action="store_true", default=False)


def add_merge_parser(sub):
    p = sub.add_parser("merge", help="Merge occ and nuc calls")
    p.add_argument("--occpeaks", required=True, help="Output from occ utility")
    p.add_argument("--nucpos", required=True, help="Output from nuc utility")
    p.add_argument("--out", help="output file basename")
    p.add_argument("--sep", default=120, help="minimum separation between call")
    p.add_argument("--min_occ", default=0.1, help="minimum lower bound occupancy of nucleosomes to be considered")


def add_nfr_parser(sub):
    p = sub.add_parser("nfr", help="Call NFRs")
    p.add_argument("--bed", required=True)
    p.add_argument("--occ_track", required=True, help="bgzip compressed, tabix-indexed bedgraph file with occupancy track")
    p.add_argument("--calls", required=True, help="bed file with nucleosome center calls")
    p.add_argument("--ins_track", help="insertion track; generated from --bam if not included")
    p.add_argument("--bam")
    p.add_argument("--fasta")
    p.add_argument("--pwm", default="Human")
    p.add_argument("--out")
    p.add_argument("--cores", type=int, default=1)
    p.add_argument("--max_occ", type=float, default=0.1)
    p.add_argument("--max_occ_upper", type=float, default=0.25)


def add_run_parser(sub):
    p = sub.add_parser("run", help="Main nucleoatac utility -- occupancy determination & calling nuc positions")
    p.add_argument("--bed", required=True)
    p.add_argument("--bam", required=True)
    p.add_argument("--out", required=True)
    p.add_argument("--fasta", required=True)
    p.add_argument("--pwm", default="Human")
    p.add_argument("--cores", type=int, default=1)
    p.add_argument("--write_all", action="store_true", default=False)


def nucleoatac_parser():
    parser = argparse.ArgumentParser(prog="nucleoatac", description="NucleoATAC on AMD MI355X")
    sub = parser.add_subparsers(dest="call")
    add_run_parser(sub)
    add_occ_parser(sub)
    add_vprocess_parser(sub)
    add_nuc_parser(sub)
    add_merge_parser(sub)
    add_nfr_parser(sub)
    return parser


def _rank0_only(fn, args):
    """vprocess / merge are single-process steps (a template, a merge of two call files): under torchrun rank 0 runs them and
    the others wait for its ok / failed flag (shard.run_on_rank0), so a failure on rank 0 ends every rank at once.  `nfr` is
    sharded across the ranks like occ and nuc (run_nfr.py)."""
    from ..shard import barrier, run_on_rank0
    run_on_rank0(fn, args)
    barrier()


def nucleoatac_main(args):
    if args.call == "occ":
        from .run_occ import run_occ
        print("---------Computing Occupancy and Nucleosomal Insert Distribution------")
        run_occ(args)
    elif args.call == "vprocess":
        from .run_vprocess import run_vprocess
        print("---------Processing VPlot---------------------------------------------")
        _rank0_only(run_vprocess, args)
    elif args.call == "nuc":
        from .run_nuc import run_nuc
        print("---------Obtaining nucleosome signal and calling positions-------------")
        run_nuc(args)
    elif args.call == "merge":
        from .merge import run_merge
        print("---------Merging------------------------------------------------------")
        _rank0_only(run_merge, args)
    elif args.call == "nfr":
        from .run_nfr import run_nfr
        print("---------Calling NFR positions----------------------------------------")
        run_nfr(args)
    elif args.call == "run":
        run_chain(args)
    else:
        raise SystemExit("usage: nucleoatac {run,occ,vprocess,nuc,merge,nfr} ...")


def run_chain(args, on_step=None):
    """`nucleoatac run`: the five steps chained through their output files exactly as the reference does (cli.py:34-64); the three
    occupancy tracks of step 1 additionally stay in HBM for steps 3 and 5 of this process (occstore.py).  on_step(name, seconds), if
    given, is called after every step (bench.py / tools/e2e_run.py read the drivers' phase clocks there)."""
    import time
    parser = nucleoatac_parser()
    occ_args = parser.parse_args(["occ", "--bed", args.bed, "--bam", args.bam, "--fasta", args.fasta, "--pwm", args.pwm,
                                  "--out", args.out, "--cores", str(args.cores)])
    vprocess_args = parser.parse_args(["vprocess", "--sizes", args.out + ".nuc_dist.txt", "--out", args.out])
    nuc_list = ["nuc", "--bed", args.bed, "--bam", args.bam, "--out", args.out, "--cores", str(args.cores), "--occ_track",
                args.out + ".occ.bedgraph.gz", "--vmat", args.out + ".VMat", "--fasta", args.fasta, "--pwm", args.pwm,
                "--sizes", args.out + ".fragmentsizes.txt"]
    if args.write_all:
        nuc_list.append("--write_all")
    nuc_args = parser.parse_args(nuc_list)
    merge_args = parser.parse_args(["merge", "--occpeaks", args.out + ".occpeaks.bed.gz", "--nucpos", args.out + ".nucpos.bed.gz",
                                    "--out", args.out])
    nfr_args = parser.parse_args(["nfr", "--bed", args.bed, "--occ_track", args.out + ".occ.bedgraph.gz", "--calls",
                                  args.out + ".nucmap_combined.bed.gz", "--out", args.out, "--fasta", args.fasta, "--pwm",
                                  args.pwm, "--bam", args.bam])
    from .merge import run_merge
    from .run_nfr import run_nfr
    from .run_nuc import run_nuc
    from .run_occ import run_occ
    from .run_vprocess import run_vprocess
    from .. import occstore
    from ..shard import barrier
    occ_args.keep_resident = True      # the occupancy tracks also stay in HBM for steps 3 and 5 of this process (occstore.py)
    steps = (("occ", "Step1: Computing Occupancy and Nucleosomal Insert Distribution---------", lambda: (run_occ(occ_args), barrier())),
             ("vprocess", "Step2: Processing Vplot------------------------------------------------", lambda: _rank0_only(run_vprocess, vprocess_args)),
             ("nuc", "Step3: Obtaining nucleosome signal and calling positions---------------", lambda: (run_nuc(nuc_args), barrier())),
             ("merge", "Step4: Making combined nucleosome position map ------------------------", lambda: _rank0_only(run_merge, merge_args)),
             ("nfr", "Step5: Calling NFR positions-------------------------------------------", lambda: (run_nfr(nfr_args), barrier())))
    try:
        for name, banner, fn in steps:
            print("---------" + banner)
            t0 = time.perf_counter()
            fn()
            if on_step is not None:
                on_step(name, time.perf_counter() - t0)
    finally:
        occstore.release(args.out + ".occ.bedgraph.gz")


def _init_distributed():
    """under torchrun (WORLD_SIZE > 1) the chunk list is sharded over the ranks; torch.distributed (gloo by default;
    NATAC_DIST_BACKEND=nccl adds a probed RCCL group) only carries the barrier and the gather of small per-chunk results"""
    from ..shard import ensure_distributed
    return ensure_distributed()[0]


def main(argv=None):
    args = nucleoatac_parser().parse_args(argv)
    dist = _init_distributed()
    try:
        nucleoatac_main(args)
    finally:
        if dist is not None and dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
