"""`nucleoatac occ | nuc` command line with the reference's flag names and defaults (nucleoatac/cli.py:92-264).
Only the sub-commands on the accelerated path exist here."""
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


def nucleoatac_parser():
    parser = argparse.ArgumentParser(prog="nucleoatac", description="NucleoATAC occ + nuc on AMD MI355X")
    sub = parser.add_subparsers(dest="call")
    add_occ_parser(sub)
    add_nuc_parser(sub)
    return parser


def nucleoatac_main(args):
    if args.call == "occ":
        from .run_occ import run_occ
        print("---------Computing Occupancy and Nucleosomal Insert Distribution------")
        run_occ(args)
    elif args.call == "nuc":
        from .run_nuc import run_nuc
        print("---------Obtaining nucleosome signal and calling positions-------------")
        run_nuc(args)
    else:
        raise SystemExit("usage: nucleoatac {occ,nuc} ...")


def _init_distributed():
    """under torchrun (WORLD_SIZE > 1) the chunk list is sharded over the ranks; torch.distributed (RCCL = backend "nccl",
    or NATAC_DIST_BACKEND=gloo) only carries the barrier and the gather of small per-chunk results"""
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
