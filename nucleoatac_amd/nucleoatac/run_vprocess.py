"""`nucleoatac vprocess` (reference: nucleoatac/run_vprocess.py:20-42): process a raw V-plot into the nucleosome template
used by `nuc`.  A one-off host step on a ~150 x 121 matrix; the plotting options of the reference are not carried over."""
from ..pyatac.fragmentsizes import FragmentSizes
from ..pyatac.VMat import VMat


def run_vprocess(args):
    vmat = VMat.open(args.vplot) if args.vplot else VMat.standard()
    vmat.mat = vmat.mat.copy()
    vmat.trim(args.lower, args.upper, args.flank)       # trim, symmetrize (run_vprocess.py:25-26)
    vmat.symmetrize()
    if args.sizes is not None:                          # insert-size normalisation (:28-31)
        vmat.norm_y(FragmentSizes.open(args.sizes))
    if args.smooth > 0:                                 # :33-34
        vmat.smooth(sd=args.smooth)
    vmat.norm()                                         # :36
    if getattr(args, "plot_extra", False):
        raise SystemExit("--plot_extra is not supported (the reference's own implementation calls VMat methods that do "
                         "not exist, run_vprocess.py:38-39)")
    vmat.save(args.out + ".VMat")
    return vmat
