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
from ..shard import balanced_ranges, barrier, ensure_distributed, env_rank_world
from ..writer import bgzip_file, tabix_index, write_bedgraph
from .NucleosomeCalling import NucParameters, nuc_batch

BATCH_CHUNKS = 4096
COMPRESS_LEVEL = 4


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
    pool = None
    if getattr(args, "cores", 1) and args.cores > 1:
        # host pool for the per-nucleosome L-BFGS fits (the reference's --cores); spawn: workers never touch the GPU
        import multiprocessing
        from concurrent.futures import ProcessPoolExecutor
        # one BLAS / OpenMP thread per worker: N processes x all-core thread pools oversubscribe the host (measured 40x slower)
        for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            os.environ[var] = "1"
        pool = ProcessPoolExecutor(max_workers=args.cores, mp_context=multiprocessing.get_context("spawn"))
        params.pool, params.pool_workers = pool, args.cores
    outputs = ["nucpos", "nucpos.redundant", "nucleoatac_signal", "nucleoatac_signal.smooth"]
    if args.write_all:
        outputs += ["nucleoatac_background", "nucleoatac_raw"]
    ensure_distributed()
    rank, world, _ = env_rank_world()
    lo, hi = balanced_ranges([c.length() for c in chunks], np.arange(len(chunks) + 1), world)[rank]
    mine = chunks[lo:hi]
    suffix = "" if world == 1 else ".rank%d" % rank
    track_keys = {"nucleoatac_signal": "norm_signal", "nucleoatac_signal.smooth": "smoothed",
                  "nucleoatac_background": "bias", "nucleoatac_raw": "nuc_signal"}
    tracks = [n for n in outputs if n in track_keys]
    paths = {n: args.out + "." + n + ".bedgraph.gz" + suffix for n in tracks}
    handles = {n: open(args.out + "." + n + ".bed" + suffix, "w") for n in outputs if n.startswith("nucpos")}
    nb = max(1, (len(mine) + BATCH_CHUNKS - 1) // BATCH_CHUNKS)
    for bi in range(nb):
        part = mine[bi * BATCH_CHUNKS:(bi + 1) * BATCH_CHUNKS]
        if not part:
            for n in tracks:
                write_bedgraph(paths[n], [], [], [0], np.zeros(0), append=bi > 0, compress=COMPRESS_LEVEL, finish=True)
            break
        try:
            nucs, flat = nuc_batch(part, params, with_flat=True)
        except Exception:
            print("Caught exception when processing:\n" + "\n".join(c.asBed() for c in part[:3]) + "\n")
            raise
        chroms, starts = [c.chrom for c in part], [c.start for c in part]
        for n in tracks:
            write_bedgraph(paths[n], chroms, starts, flat["out_off"], flat[track_keys[n]], append=bi > 0,
                           compress=COMPRESS_LEVEL, finish=(bi == nb - 1 and rank == world - 1))
        for nuc in nucs:
            for i in sorted(nuc.nonredundant):
                nuc.nuc_collection[i].write(handles["nucpos"])
            for i in sorted(nuc.redundant):
                nuc.nuc_collection[i].write(handles["nucpos.redundant"])
            nuc.removeData()
    for h in handles.values():
        h.close()
    if pool is not None:
        pool.shutdown()
    barrier()      # every rank has closed its part files (raises if WORLD_SIZE > 1 without a process group)
    if rank == 0:
        for n in outputs:
            base = args.out + "." + n + (".bed" if n.startswith("nucpos") else ".bedgraph.gz")
            if world > 1:
                with open(base, "wb") as fo:
                    for r in range(world):
                        with open(base + ".rank%d" % r, "rb") as fi:
                            shutil.copyfileobj(fi, fo)
                        os.remove(base + ".rank%d" % r)
            # bgzip + tabix of every output like the reference (run_nuc.py:195-201)
            if n.startswith("nucpos"):
                bgzip_file(base, level=COMPRESS_LEVEL)
                tabix_index(base + ".gz")
            else:
                tabix_index(base)
