"""`nucleoatac nuc` (reference: nucleoatac/run_nuc.py:141-201): NucleoATAC signal tracks + nucleosome calls."""
import gzip
import os
import shutil

import numpy as np

from ..pyatac.bias import PWM
from ..pyatac.chunk import ChunkList
from ..pyatac.fragmentsizes import FragmentSizes
from ..pyatac.utils import read_chrom_sizes_from_bam, read_chrom_sizes_from_fasta
from ..pyatac.VMat import VMat
from ..shard import balanced_ranges, env_rank_world
from .NucleosomeCalling import NucParameters, nuc_batch

BATCH_CHUNKS = 4096


def _nucHelper(arg):
    """dict of per-chunk outputs, same keys as the reference's helper (run_nuc.py:22-39)"""
    chunk, params = arg
    return _nucHelperBatch([chunk], params)[0]


def _nucHelperBatch(chunks, params):
    out = []
    try:
        for nuc in nuc_batch(chunks, params):
            out.append({"nucpos": [nuc.nuc_collection[i] for i in sorted(nuc.nonredundant)],
                        "nucpos.redundant": [nuc.nuc_collection[i] for i in sorted(nuc.redundant)],
                        "nucleoatac_signal": nuc.norm_signal, "nucleoatac_raw": nuc.nuc_signal,
                        "nucleoatac_background": nuc.bias, "nucleoatac_signal.smooth": nuc.smoothed})
            nuc.removeData()
    except Exception:
        print("Caught exception when processing:\n" + "\n".join(c.asBed() for c in chunks[:3]) + "\n")
        raise
    return out


def run_nuc(args):
    vmat = VMat.open(args.vmat)
    chrs = read_chrom_sizes_from_fasta(args.fasta) if args.fasta else read_chrom_sizes_from_bam(args.bam)
    pwm = PWM.open(args.pwm)
    chunks = ChunkList.read(args.bed, chromDict=chrs,
                            min_offset=vmat.mat.shape[1] + vmat.upper // 2 + max(pwm.up, pwm.down) + args.nuc_sep // 2,
                            min_length=args.nuc_sep * 2)
    chunks.slop(chrs, up=args.nuc_sep // 2, down=args.nuc_sep // 2)
    chunks.merge()
    if args.sizes is not None:
        fragment_dist = FragmentSizes.open(args.sizes)
    else:
        fragment_dist = FragmentSizes(0, upper=vmat.upper)
        fragment_dist.calculateSizes(args.bam, chunks)
    params = NucParameters(vmat=vmat, fragmentsizes=fragment_dist, bam=args.bam, fasta=args.fasta, pwm=args.pwm,
                           occ_track=args.occ_track, sd=args.sd, nonredundant_sep=args.nuc_sep,
                           redundant_sep=args.redundant_sep, min_z=args.min_z, min_lr=args.min_lr, atac=args.atac)
    outputs = ["nucpos", "nucpos.redundant", "nucleoatac_signal", "nucleoatac_signal.smooth"]
    if args.write_all:
        outputs += ["nucleoatac_background", "nucleoatac_raw"]
    rank, world, _ = env_rank_world()
    lo, hi = balanced_ranges([c.length() for c in chunks], np.arange(len(chunks) + 1), world)[rank]
    mine = chunks[lo:hi]
    suffix = "" if world == 1 else ".rank%d" % rank
    ext = lambda n: ".bed" if n.startswith("nucpos") else ".bedgraph"
    handles = {n: open(args.out + "." + n + ext(n) + suffix, "w") for n in outputs}
    for i in range(0, len(mine), BATCH_CHUNKS):
        for res in _nucHelperBatch(mine[i:i + BATCH_CHUNKS], params):
            for n in outputs:
                if n.startswith("nucpos"):
                    for pos in res[n]:
                        pos.write(handles[n])
                else:
                    res[n].write_track(handles[n])
    for h in handles.values():
        h.close()
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
    if rank == 0:
        for n in outputs:
            base = args.out + "." + n + ext(n)
            if world > 1:
                with open(base, "w") as fo:
                    for r in range(world):
                        with open(base + ".rank%d" % r) as fi:
                            shutil.copyfileobj(fi, fo)
                        os.remove(base + ".rank%d" % r)
            with open(base, "rb") as fi, gzip.open(base + ".gz", "wb") as fo:
                shutil.copyfileobj(fi, fo)
            os.remove(base)
