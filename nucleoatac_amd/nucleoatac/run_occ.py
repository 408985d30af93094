"""`nucleoatac occ` (reference: nucleoatac/run_occ.py:77-148): occupancy tracks + peaks + nucleosomal size
distribution.  The per-chunk Pool.map of the reference is replaced by GPU batches; with torchrun / WORLD_SIZE > 1
the chunk list is sharded across GPUs (nucleoatac_amd/shard.py) and rank r writes `<out>.rank<r>.*` part files
that rank 0 concatenates in chunk order."""
import gzip
import os
import shutil

import numpy as np

from ..pyatac.bias import PWM
from ..pyatac.chunk import ChunkList
from ..pyatac.fragmentsizes import FragmentSizes
from ..pyatac.utils import read_chrom_sizes_from_bam, read_chrom_sizes_from_fasta
from ..shard import balanced_ranges, env_rank_world, gather_in_chunk_order, ordered_sum
from .Occupancy import FragmentMixDistribution, OccupancyParameters, occ_batch

BATCH_CHUNKS = 4096   # chunks per GPU batch (the reference maps cores*5 chunks per pool.map round)


def _occHelper(arg):
    """(nuc_dist, OccupancyTrack, [OccPeak]) for one chunk -- same return shape as the reference's helper
    (run_occ.py:23-39); `_occHelperBatch` is what the driver uses"""
    chunk, params = arg
    return _occHelperBatch([chunk], params)[0]


def _occHelperBatch(chunks, params):
    out = []
    try:
        for oc in occ_batch(chunks, params):
            out.append((oc.getNucDist(), oc.occ, [oc.peaks[i] for i in sorted(oc.peaks.keys())]))
            oc.removeData()
    except Exception:
        print("Caught exception when processing:\n" + "\n".join(c.asBed() for c in chunks[:3]) + "\n")
        raise
    return out


def _compress(path):
    with open(path, "rb") as fi, gzip.open(path + ".gz", "wb") as fo:
        shutil.copyfileobj(fi, fo)
    os.remove(path)


def run_occ(args):
    chrs = read_chrom_sizes_from_fasta(args.fasta) if args.fasta else read_chrom_sizes_from_bam(args.bam)
    pwm = PWM.open(args.pwm)
    chunks = ChunkList.read(args.bed, chromDict=chrs,
                            min_offset=args.flank + args.upper // 2 + max(pwm.up, pwm.down) + args.nuc_sep // 2)
    chunks.slop(chrs, up=args.nuc_sep // 2, down=args.nuc_sep // 2)
    chunks.merge()
    fragment_dist = FragmentMixDistribution(0, upper=args.upper)
    if args.sizes is not None:
        tmp = FragmentSizes.open(args.sizes)
        fragment_dist.fragmentsizes = FragmentSizes(0, args.upper, vals=tmp.get(0, args.upper))
    else:
        fragment_dist.getFragmentSizes(args.bam, chunks)
    fragment_dist.modelNFR()
    rank, world, _ = env_rank_world()
    if rank == 0:
        fragment_dist.fragmentsizes.save(args.out + ".fragmentsizes.txt")
    params = OccupancyParameters(fragment_dist, args.upper, args.fasta, args.pwm, sep=args.nuc_sep, min_occ=args.min_occ,
                                 flank=args.flank, bam=args.bam, ci=args.confidence_interval, step=args.step)
    from ..pyatac.fragments import FragmentStore
    st = FragmentStore.open(args.bam)
    nfr_per_chunk = [len(st.fetch(c.chrom, c.start, c.end)[0]) for c in chunks]
    lo, hi = balanced_ranges([c.length() for c in chunks], np.concatenate(([0], np.cumsum(nfr_per_chunk))), world)[rank]
    mine = chunks[lo:hi]
    suffix = "" if world == 1 else ".rank%d" % rank
    names = ("occ", "occ.lower_bound", "occ.upper_bound")
    handles = [open(args.out + "." + n + ".bedgraph" + suffix, "w") for n in names]
    peaks_handle = open(args.out + ".occpeaks.bed" + suffix, "w")
    dists = []
    for i in range(0, len(mine), BATCH_CHUNKS):
        for nuc_dist, track, peaks in _occHelperBatch(mine[i:i + BATCH_CHUNKS], params):
            dists.append(nuc_dist)
            track.write_track(handles[0], vals=track.smoothed_vals)
            track.write_track(handles[1], vals=track.smoothed_lower)
            track.write_track(handles[2], vals=track.smoothed_upper)
            for p in peaks:
                p.write(peaks_handle)
    for h in handles + [peaks_handle]:
        h.close()
    dists = gather_in_chunk_order(dists, dst=0)
    if world > 1:
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
    if rank == 0:
        for n in list(names) + ["occpeaks"]:
            ext = ".bed" if n == "occpeaks" else ".bedgraph"
            base = args.out + "." + n + ext
            if world > 1:
                with open(base, "w") as fo:
                    for r in range(world):
                        with open(base + ".rank%d" % r) as fi:
                            shutil.copyfileobj(fi, fo)
                        os.remove(base + ".rank%d" % r)
            _compress(base)
        nuc_dist = ordered_sum(dists) if dists else np.zeros(args.upper)
        FragmentSizes(0, args.upper, vals=nuc_dist).save(args.out + ".nuc_dist.txt")
