"""`nucleoatac nfr` (reference: nucleoatac/run_nfr.py:71-130): NFR positions between the combined nucleosome calls, plus the
insertion track of the regions when none is given.  The reference's pool.map over chunks becomes one GPU batch per
BATCH_CHUNKS chunks for the insertion counts (natac_run_ins) and one native multi-threaded writer call per batch.  With
torchrun / WORLD_SIZE > 1 the chunk list is sharded across the ranks like `occ` and `nuc` (rank part files, concatenated by
rank 0 in chunk order): the per-chunk loop is host work -- two tabix reads, one PWM launch -- and would otherwise keep every
other rank waiting in a barrier for as long as one process needs for the whole genome."""
import os
import shutil

import numpy as np

from ..pyatac.bias import PWM
from ..pyatac.chunk import ChunkList
from ..pyatac.utils import read_chrom_sizes_from_bam, read_chrom_sizes_from_fasta
from ..shard import barrier, ensure_distributed, env_rank_world
from ..writer import bgzip_file, tabix_index, write_bed_rows, write_bedgraph
from .NFRCalling import NFRChunk, NFRParameters, nfr_batch

BATCH_CHUNKS = 4096
COMPRESS_LEVEL = 4
LAST_TIMINGS = {}      # phase -> seconds of the last run_nfr call of this process (bench.py's cli_end_to_end reports them)


def _nfrHelper(arg):
    """(nfrs, ins) or nfrs for one chunk -- the reference's helper (run_nfr.py:21-39)"""
    chunk, params = arg
    nfr = NFRChunk(chunk)
    nfr.process(params)
    out = (nfr.nfrs, nfr.ins) if params.ins_track is None else nfr.nfrs
    return out


def _batch_insertions(chunks, bam, as_members=False):
    """InsertionTrack.calculateInsertions (lower 0, upper 2000, tracks.py:164-168) of every chunk in one GPU batch; with
    `as_members` also the track as finished BGZF members + its tabix records (Track.write_track + bgzip on the device)"""
    from .. import _lib as L
    from .. import get_context
    from ..pipeline import pack
    pk = pack(chunks, bam)
    b = get_context().upload(pk)
    z = info = None
    try:
        b.run_ins(0, 2000)
        flat = b.track(L.T_INS).astype(np.float64)
        if as_members:
            z, info = b.format_track(L.T_INS, [c.chrom for c in chunks], [c.start for c in chunks], compress=True)
            if info["hard"]:
                z = info = None
    finally:
        b.free()
    return pk.out_off, flat, z, info


def run_nfr(args):
    from .run_occ import _Phases
    ph = _Phases(LAST_TIMINGS)
    if args.bam is None and args.ins_track is None:
        raise Exception("Must supply either bam file or insertion track")
    if not args.out:
        args.out = ".".join(os.path.basename(args.calls).split(".")[0:-3])
    if env_rank_world()[2] == 0 and isinstance(args.bam, str):      # the node's publishing rank
        from ..pyatac.fragments import FragmentStore
        FragmentStore.prefetch(args.bam)       # it decodes (shard.shared_fragment_store): start now, next to the FASTA index / BED reads
    if getattr(args, "fasta", None):
        from ..pyatac.seq import FastaStore
        FastaStore.prefetch(args.fasta)        # the genome loads on its own thread; the BED file only needs the record lengths
    if args.fasta is not None:
        chrs_fasta = read_chrom_sizes_from_fasta(args.fasta)
        pwm = PWM.open(args.pwm)
        chunks = ChunkList.read(args.bed, chromDict=chrs_fasta, min_offset=max(pwm.up, pwm.down))
    else:
        import warnings
        warnings.warn("nfr without --fasta: the bias column of nfrpos.bed is written as 'nan' (the reference raises here; see "
                      "INTEGRATION.md section 6)")
        chunks = ChunkList.read(args.bed)
    ensure_distributed()
    if args.bam is not None:
        from ..shard import shared_fragment_store
        shared_fragment_store(args.bam)      # decoded once per node; FragmentStore.open(args.bam) returns it on every rank
        chunks.checkChroms(read_chrom_sizes_from_bam(args.bam), chrom_source="BAM file")
    chunks.merge()
    params = NFRParameters(args.occ_track, args.calls, args.ins_track, args.bam, max_occ=args.max_occ,
                           max_occ_upper=args.max_occ_upper, fasta=args.fasta, pwm=args.pwm)
    make_ins = params.ins_track is None
    rank, world, _ = env_rank_world()
    lens = np.array([c.length() for c in chunks], dtype=np.float64)
    cum = np.concatenate(([0.0], np.cumsum(lens)))
    cuts = [int(np.searchsorted(cum, cum[-1] * r / world, "left")) for r in range(world)] + [len(chunks)]
    chunks = chunks[cuts[rank]:max(cuts[rank], cuts[rank + 1])]
    suffix = "" if world == 1 else ".rank%d" % rank
    ins_path = args.out + ".ins.bedgraph.gz" + suffix
    nb = max(1, (len(chunks) + BATCH_CHUNKS - 1) // BATCH_CHUNKS)
    nfr_path = args.out + ".nfrpos.bed" + suffix
    open(nfr_path, "w").close()
    # one rank: the insertion track leaves the GPU as BGZF members with its tabix records (no formatting on the host, the file is
    # not read back for the index); several ranks write part files through the host writer, rank 0 indexes the concatenation
    from ..writer import BGZF_EOF, TbiBuilder
    from .run_occ import DEVICE_WRITER
    on_device = make_ins and world == 1 and DEVICE_WRITER
    tbi = TbiBuilder() if on_device else None
    ins_bytes = 0
    ph.mark("read_inputs")
    for bi in range(nb):
        part = chunks[bi * BATCH_CHUNKS:(bi + 1) * BATCH_CHUNKS]
        if not part:
            if tbi is not None:
                tbi.close()
                tbi = None
            if make_ins:
                write_bedgraph(ins_path, [], [], [0], np.zeros(0), append=bi > 0, compress=COMPRESS_LEVEL,
                               finish=(rank == world - 1))
            break
        off = flat = z = zinfo = None
        if make_ins:
            off, flat, z, zinfo = _batch_insertions(part, args.bam, as_members=on_device)
            ph.mark("insertions_gpu")
        try:
            kc, left, right, vals = nfr_batch(part, params, off, flat)
        except Exception:
            print("Caught exception when processing:\n" + "\n".join(c.asBed() for c in part[:3]) + "\n")
            raise
        ph.mark("reads_bias_statistics")
        if len(kc):        # NFR.asBed rows of the whole sub-batch, python-2 float text, natively
            names = sorted(set(c.chrom for c in part))
            idx = {c: i for i, c in enumerate(names)}
            write_bed_rows(nfr_path, names, np.array([idx[c.chrom] for c in part], dtype=np.int32)[kc], left, right, vals)
        if make_ins and z is not None and tbi is not None:      # members from the device: append, log the tabix records
            with open(ins_path, "ab" if bi > 0 else "wb") as fh:
                fh.write(memoryview(z))
                if bi == nb - 1:
                    fh.write(BGZF_EOF)
            tbi.push(zinfo["index"], ins_bytes)
            ins_bytes += len(z)
        elif make_ins:    # Track.write_track of every chunk's insertion track (run_nfr.py:55-67) through the native writer
            if tbi is not None:           # a sub-batch the device could not format: this file gets its index from the file
                tbi.close()
                tbi = None
            write_bedgraph(ins_path, [c.chrom for c in part], [c.start for c in part], off, flat, append=bi > 0,
                           compress=COMPRESS_LEVEL, finish=(bi == nb - 1 and rank == world - 1))
        ph.mark("write_rows_and_ins_track")
    barrier()          # every rank has closed its part files
    if rank != 0:
        return
    if world > 1:      # text lines / BGZF members concatenate: rank order == chunk order
        for base in [args.out + ".nfrpos.bed"] + ([args.out + ".ins.bedgraph.gz"] if make_ins else []):
            with open(base, "wb") as fo:
                for r in range(world):
                    with open(base + ".rank%d" % r, "rb") as fi:
                        shutil.copyfileobj(fi, fo)
                    os.remove(base + ".rank%d" % r)
        ins_path = args.out + ".ins.bedgraph.gz"
    bgzip_file(args.out + ".nfrpos.bed", level=COMPRESS_LEVEL)       # pysam.tabix_compress + tabix_index (run_nfr.py:121-128)
    tabix_index(args.out + ".nfrpos.bed.gz")
    if make_ins and tbi is not None:
        tbi.write(ins_path + ".tbi")
        tbi.close()
    elif make_ins:
        tabix_index(ins_path)
    ph.mark("bgzip_tabix")
