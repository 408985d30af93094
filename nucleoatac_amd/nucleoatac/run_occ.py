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
from ..shard import balanced_ranges, barrier, ensure_distributed, env_rank_world, gather_in_chunk_order, ordered_sum
from ..writer import bgzip_file, tabix_index, write_bedgraph
from .Occupancy import FragmentMixDistribution, OccupancyParameters, occ_batch

BATCH_CHUNKS = 4096   # chunks per GPU batch (the reference maps cores*5 chunks per pool.map round)
COMPRESS_LEVEL = 4     # BGZF deflate level of the track files


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
    ensure_distributed()
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
    names = {"occ": "smoothed_vals", "occ.lower_bound": "smoothed_lower", "occ.upper_bound": "smoothed_upper"}
    paths = {n: args.out + "." + n + ".bedgraph.gz" + suffix for n in names}
    peaks_handle = open(args.out + ".occpeaks.bed" + suffix, "w")
    dists = []
    nb = max(1, (len(mine) + BATCH_CHUNKS - 1) // BATCH_CHUNKS)
    for bi in range(nb):
        part = mine[bi * BATCH_CHUNKS:(bi + 1) * BATCH_CHUNKS]
        if not part:
            for n in names:
                write_bedgraph(paths[n], [], [], [0], np.zeros(0), append=bi > 0, compress=COMPRESS_LEVEL, finish=True)
            break
        try:
            occs, flat = occ_batch(part, params, with_flat=True)
        except Exception:
            print("Caught exception when processing:\n" + "\n".join(c.asBed() for c in part[:3]) + "\n")
            raise
        chroms, starts = [c.chrom for c in part], [c.start for c in part]
        for n, key in names.items():     # native multi-threaded run-length writer + BGZF (tracks.py:37-74, run_occ.py:130-136)
            write_bedgraph(paths[n], chroms, starts, flat["out_off"], flat[key], append=bi > 0, compress=COMPRESS_LEVEL,
                           finish=(bi == nb - 1 and rank == world - 1))
        for oc in occs:
            dists.append(oc.getNucDist())
            for i in sorted(oc.peaks.keys()):
                oc.peaks[i].write(peaks_handle)
            oc.removeData()
    peaks_handle.close()
    dists = gather_in_chunk_order(dists, dst=0)
    barrier()      # every rank has closed its part files (raises if WORLD_SIZE > 1 without a process group)
    if rank == 0:
        if world > 1:   # BGZF members / text lines concatenate: rank order == chunk order
            for n in list(names) + ["occpeaks"]:
                base = args.out + "." + n + (".bed" if n == "occpeaks" else ".bedgraph.gz")
                with open(base, "wb") as fo:
                    for r in range(world):
                        with open(base + ".rank%d" % r, "rb") as fi:
                            shutil.copyfileobj(fi, fo)
                        os.remove(base + ".rank%d" % r)
        # bgzip + tabix of every output like the reference (run_occ.py:130-136)
        bgzip_file(args.out + ".occpeaks.bed", level=COMPRESS_LEVEL)
        tabix_index(args.out + ".occpeaks.bed.gz")
        for n in names:
            tabix_index(args.out + "." + n + ".bedgraph.gz")
        nuc_dist = ordered_sum(dists) if dists else np.zeros(args.upper)
        FragmentSizes(0, args.upper, vals=nuc_dist).save(args.out + ".nuc_dist.txt")
