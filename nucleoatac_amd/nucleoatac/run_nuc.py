"""`nucleoatac nuc` (reference: nucleoatac/run_nuc.py:141-201): NucleoATAC signal tracks + nucleosome calls.

Like `occ` (run_occ.py) the chunk list goes through `nucleoatac_amd.executor.PipelinedExecutor`; a writer thread consumes the
finished sub-batches in chunk order -- native run-length bedGraph + BGZF for the tracks, then the calls: the candidate
statistics (LR, z) come from the device, the thresholds of `findAllNucs` (NucleosomeCalling.py:303-311) are applied to whole
sub-batches with numpy, `reduce_peaks` and the per-nucleosome L-BFGS fuzziness fit (NucleosomeCalling.py:137-194, host by
SURVEY.md section 8f row 3) run per chunk, the fits on the `--cores` process pool while the GPU works on the next sub-batches."""
import os
import shutil

import numpy as np

from .. import _lib as L
from ..executor import PipelinedExecutor, Stages
from ..pipeline import pack, prefetch_map, sub_batches
from ..pyatac.bias import PWM
from ..pyatac.chunk import ChunkList
from ..pyatac.fragmentsizes import FragmentSizes
from ..pyatac.utils import read_chrom_sizes_from_bam, read_chrom_sizes_from_fasta, reduce_peaks
from ..pyatac.VMat import VMat
from ..shard import balanced_ranges, barrier, broadcast_object, ensure_distributed, env_rank_world, shared_fragment_store
from ..writer import bgzip_file, tabix_index, write_bed_rows, write_bedgraph
from .NucleosomeCalling import NucParameters, fit_fuzz_tasks, nuc_batch, read_occ_tracks_many
from .run_occ import DEVICE_WRITER, _Phases, _Writer, finish_indexes

LAST_TIMINGS = {}

BATCH_CHUNKS = int(os.environ.get("NATAC_BATCH_CHUNKS", "4096"))
# bases per sub-batch of the nuc pipeline.  The writer finishes sub-batch k (waits for its fits) only after it has started sub-batch
# k + 1 (batch_calls_start / _finish), so the fit pool does not idle at sub-batch borders and small sub-batches cost nothing there
# while the first result arrives sooner.  `nucleoatac run` on 60 k x 10 kb tiles, seconds of `nuc`: 4,096 chunks = 41 Mbp without the
# look-ahead 21.5, with it 17.1; 18 Mbp 14.7; 9 Mbp 14.4; 4.5 Mbp 15.6 (2-kb windows: 4,096 chunks are 8.7 Mbp either way).
NUC_SUB_BP = int(os.environ.get("NATAC_NUC_SUB_BP", "9000000"))
N_CONTEXTS = int(os.environ.get("NATAC_CONTEXTS", "3"))
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


def batch_calls_start(r, params, pool=None, pool_workers=1):
    """first half of batch_calls: everything that needs the sub-batch's downloaded arrays -- thresholds, values at the calls, the
    fuzziness-fit tasks (SUBMITTED to the pool here), the occupancy values -- so that the result's page-locked buffers can go back to
    the executor; batch_calls_finish(state) waits for the fits and forms the rows.  The writer finishes sub-batch k only after it has
    started sub-batch k + 1: the fit pool always holds the next sub-batch's tasks when it runs out of this one's."""
    part, pk = r.tag, r.packed
    cc, cp, lr, _var, z = r.peaks
    tr = r.tracks
    over = set(int(k) for k in np.nonzero(r.status & 2)[0])
    idx = pk.out_off[cc] + cp
    nuc_cov = tr[L.T_NUC_COV][idx]
    with np.errstate(invalid="ignore"):
        keep = (nuc_cov > params.min_reads) & (lr > params.min_lr) & (z >= params.min_z)     # the reference's order: reads, LR, z
    if over:
        keep &= ~np.isin(cc, list(over))
    kc, kp, kidx = cc[keep], cp[keep], idx[keep]
    vals = np.empty((len(kc), 10), dtype=np.float64)
    vals[:, 0], vals[:, 4] = z[keep], lr[keep]
    vals[:, 1:4] = np.nan
    vals[:, 5], vals[:, 6] = tr[L.T_NORM][kidx], tr[L.T_RAW][kidx]
    vals[:, 7], vals[:, 8] = nuc_cov[keep], tr[L.T_NFR_COV][kidx]
    bounds = np.searchsorted(kc, np.arange(len(part) + 1))
    called = [k for k in range(len(part)) if bounds[k + 1] > bounds[k]]
    # fuzziness fits: one task per chunk with calls (the smoothed values are copied out of the pinned slot), started on the pool
    sm = tr[L.T_SMOOTH]
    tasks = [(sm[int(pk.out_off[k]):int(pk.out_off[k + 1])].copy(), kp[int(bounds[k]):int(bounds[k + 1])].astype(np.int64),
              params.nonredundant_sep, params.smooth_sd) for k in called]
    fits = fit_fuzz_tasks(tasks, pool, pool_workers, start_only=True)
    if params.occ_track is not None and called:
        # meanwhile the three occupancy tracks of every chunk with calls (NucChunk.getOcc, NucleosomeCalling.py:284-293): one native
        # call per file, or out of HBM when this process wrote them (occstore.py)
        for k, res in zip(called, read_occ_tracks_many(params.occ_track, [part[k] for k in called])):
            if res is not None:
                a, e = int(bounds[k]), int(bounds[k + 1])
                for j in range(3):
                    vals[a:e, 1 + j] = res[j][kp[a:e]]
    return dict(part=part, kc=kc, kp=kp, vals=vals, bounds=bounds, called=called, fits=fits, over=over, params=params)


def batch_calls_finish(st):
    """second half of batch_calls: wait for the fits of the sub-batch, reduce to the non-redundant set, merge the overflow chunks"""
    part, kc, kp, vals, bounds, called, over, params = (st[k] for k in ("part", "kc", "kp", "vals", "bounds", "called", "over", "params"))
    fits = st["fits"]()
    nonred = np.zeros(len(kc), dtype=bool)
    for k, f in zip(called, fits):
        a, e = int(bounds[k]), int(bounds[k + 1])
        vals[a:e, 9] = [x[0] for x in f]
        keys = kp[a:e]
        nr = reduce_peaks(keys, list(vals[a:e, 0]), params.nonredundant_sep)
        nonred[a:e] = np.isin(keys, nr)
    out = {"nucpos": (kc[nonred], kp[nonred], vals[nonred]), "nucpos.redundant": (kc[~nonred], kp[~nonred], vals[~nonred])}
    if over:     # chunks with more local maxima than the device peak finder holds per chunk: the per-chunk API path, merged in order
        from .. import context_lock
        for k in sorted(over):
            with context_lock:           # this runs on the writer thread, on the process-wide context (see nucleoatac_amd/__init__.py)
                nc = nuc_batch([part[k]], params)[0]
            for name, ids in (("nucpos", nc.nonredundant), ("nucpos.redundant", nc.redundant)):
                rows = [nc.nuc_collection[int(i)] for i in sorted(ids)]
                if not rows:
                    continue
                c0, p0, v0 = out[name]
                at = int(np.searchsorted(c0, k))
                add = np.array([[n.z, n.occ, n.occ_lower, n.occ_upper, n.lr, n.norm_signal, n.nuc_signal, n.nuc_cov, n.nfr_cov, n.fuzz]
                                for n in rows], dtype=np.float64)
                where = np.full(len(rows), at)
                out[name] = (np.insert(c0, where, k), np.insert(p0, where, [n.start - part[k].start for n in rows]),
                             np.insert(v0, where, add, axis=0))
    return out


def batch_calls(r, params, pool=None, pool_workers=1):
    """NucChunk.findAllNucs + fit for every chunk of a finished sub-batch (NucleosomeCalling.py:294-324) from the device's
    candidate arrays: returns {"nucpos": rows, "nucpos.redundant": rows} with rows = (chunk index, position, 10 value columns:
    z, occ, occ_lower, occ_upper, lr, norm_signal, nuc_signal, nuc_cov, nfr_cov, fuzz) in chunk / position order."""
    return batch_calls_finish(batch_calls_start(r, params, pool, pool_workers))


def run_nuc(args):
    ph = _Phases(LAST_TIMINGS)
    if env_rank_world()[2] == 0 and isinstance(args.bam, str):      # the node's publishing rank
        from ..pyatac.fragments import FragmentStore
        FragmentStore.prefetch(args.bam)       # it decodes (shard.shared_fragment_store): start now, next to the FASTA index / BED reads
    if getattr(args, "fasta", None):
        from ..pyatac.seq import FastaStore
        FastaStore.prefetch(args.fasta)        # the genome loads on its own thread; the BED file only needs the record lengths
    vmat = VMat.open(args.vmat)
    chrs = read_chrom_sizes_from_fasta(args.fasta) if args.fasta else read_chrom_sizes_from_bam(args.bam)
    pwm = PWM.open(args.pwm)
    chunks = ChunkList.read(args.bed, chromDict=chrs,
                            min_offset=vmat.mat.shape[1] + vmat.upper // 2 + max(pwm.up, pwm.down) + args.nuc_sep // 2,
                            min_length=args.nuc_sep * 2)
    chunks.slop(chrs, up=args.nuc_sep // 2, down=args.nuc_sep // 2)
    chunks.merge()
    ensure_distributed()
    rank, world, _ = env_rank_world()
    st = shared_fragment_store(args.bam)
    fragment_dist = None
    if rank == 0:           # global pre-step once
        if args.sizes is not None:
            fragment_dist = FragmentSizes.open(args.sizes)
        else:
            fragment_dist = FragmentSizes(0, upper=vmat.upper)
            fragment_dist.calculateSizes(st, chunks)
    fragment_dist = broadcast_object(fragment_dist)
    ph.mark("read_inputs_sizes")
    params = NucParameters(vmat=vmat, fragmentsizes=fragment_dist, bam=st, fasta=args.fasta, pwm=args.pwm,
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
    lo, hi = balanced_ranges([c.length() for c in chunks], np.arange(len(chunks) + 1), world)[rank]
    mine = chunks[lo:hi]
    suffix = "" if world == 1 else ".rank%d" % rank
    track_of = {"nucleoatac_signal": L.T_NORM, "nucleoatac_signal.smooth": L.T_SMOOTH, "nucleoatac_background": L.T_BACKGROUND,
                "nucleoatac_raw": L.T_RAW}
    track_of = {n: t for n, t in track_of.items() if n in outputs}
    paths = {n: args.out + "." + n + ".bedgraph.gz" + suffix for n in track_of}
    call_paths = {n: args.out + "." + n + ".bed" + suffix for n in outputs if n.startswith("nucpos")}
    for p in call_paths.values():
        open(p, "w").close()
    # sub-batches: <= BATCH_CHUNKS chunks and NUC_SUB_BP bases (pipeline.sub_batches)
    parts = sub_batches(mine, BATCH_CHUNKS, NUC_SUB_BP if "NATAC_BATCH_CHUNKS" not in os.environ else 1 << 62)
    if not parts:
        for n in track_of:
            write_bedgraph(paths[n], [], [], [0], np.zeros(0), append=False, compress=COMPRESS_LEVEL, finish=(rank == world - 1))

    calls_s = [0.0]

    def timed(fn):
        def run(x):
            import time
            t0 = time.perf_counter()
            try:
                return fn(x)
            finally:
                calls_s[0] += time.perf_counter() - t0
        return run

    def calls_start(r):          # thresholds, values at the calls, fits submitted, occupancy values: needs the result's buffers
        return batch_calls_start(r, params, pool, getattr(args, "cores", 1) or 1)

    def calls_finish(st):        # waits for the fits; rows in chunk / position order
        part = st["part"]
        names = sorted(set(c.chrom for c in part))
        idx = {c: i for i, c in enumerate(names)}
        cid_of = np.array([idx[c.chrom] for c in part], dtype=np.int32)
        start_of = np.array([c.start for c in part], dtype=np.int64)
        for name, (kc, kp, vals) in batch_calls_finish(st).items():
            if len(kc):
                pos = start_of[kc] + kp
                write_bed_rows(call_paths[name], names, cid_of[kc], pos, pos + 1, vals)

    calls = (timed(calls_start), timed(calls_finish))

    try:
      if parts:
        # the calls need coverage, raw and smoothed values at the candidates: downloaded with the tracks that are written
        need = [L.T_NORM, L.T_SMOOTH, L.T_RAW, L.T_NUC_COV, L.T_NFR_COV]
        if not DEVICE_WRITER:
            need = list(dict.fromkeys(list(track_of.values()) + need))
        stages = Stages(nuc_sd=params.smooth_sd, occ=False, ins=None,
                        peaks=dict(min_signal=0, sep=params.redundant_sep, boundary=params.nonredundant_sep // 2,
                                   order=params.redundant_sep // 2), tracks=need,
                        text_tracks=tuple(track_of.values()) if DEVICE_WRITER else ())
        writer = _Writer(paths, track_of, calls, len(parts), rank == world - 1)
        writer.start()
        fa_chrs = read_chrom_sizes_from_fasta(params.fasta) if params.fasta is not None else params.chrs

        def items():       # sub-batches packed up to three ahead of the GPU, on their own threads
            return prefetch_map(lambda part: (pack(part, st, params.fasta, fa_chrs, params.pwm, atac=params.atac, window=params.window,
                                                   upper=params.upper, bias_on_device=True), part), parts, depth=3)

        from .. import default_device
        device = default_device()
        try:
            with PipelinedExecutor(device, params.install, stages, n_contexts=min(N_CONTEXTS, len(parts))) as ex:
                for r in ex.map(items()):
                    writer.put(r)
        except Exception:
            print("Caught exception when processing:\n" + "\n".join(c.asBed() for c in mine[:3]) + "\n")
            raise
        finally:
            writer.finish()
        ph.mark("pipeline_wall")
        LAST_TIMINGS["writer_inside_pipeline"] = round(writer.seconds, 3)
        LAST_TIMINGS["calls_and_fits_inside_writer"] = round(calls_s[0], 3)
    finally:
        if pool is not None:             # also on a failure: the spawn workers of the fit pool must not outlive the run
            pool.shutdown()
    to_index = finish_indexes(writer if parts else None, list(track_of), lambda n: args.out + "." + n + ".bedgraph.gz")
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
            elif base in to_index:
                tabix_index(base)
    ph.mark("merge_bgzip_tabix")
