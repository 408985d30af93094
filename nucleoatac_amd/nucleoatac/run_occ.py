"""`nucleoatac occ` (reference: nucleoatac/run_occ.py:77-148): occupancy tracks + peaks + nucleosomal size
distribution.

The reference maps `_occHelper` over the chunks with a process pool and hands every result to writer processes through
JoinableQueues, so computing and writing overlap (run_occ.py:101-123).  Here the chunk list goes through
`nucleoatac_amd.executor.PipelinedExecutor`: sub-batches of thousands of chunks are packed on the host, uploaded, computed and
downloaded by several contexts of one GPU in turn, and a writer thread formats + BGZF-compresses the finished sub-batches in
chunk order while the next ones are still on the device.  With torchrun / WORLD_SIZE > 1 the chunk list is sharded across
GPUs (nucleoatac_amd/shard.py) and rank r writes `<out>.*.rank<r>` part files that rank 0 concatenates in chunk order; the
global pre-steps (BAM decode, insert-size histogram, modelNFR) run once, on rank 0."""
import os
import queue
import shutil
import threading

import numpy as np

from .. import _lib as L
from ..executor import PipelinedExecutor, Stages
from ..pipeline import SUB_BATCH_BP, chunk_fragment_counts, pack, prefetch_map, sub_batches
from ..pyatac.bias import PWM
from ..pyatac.chunk import ChunkList
from ..pyatac.fragmentsizes import FragmentSizes
from ..pyatac.utils import read_chrom_sizes_from_bam, read_chrom_sizes_from_fasta
from ..shard import (balanced_ranges, barrier, broadcast_object, ensure_distributed, env_rank_world, gather_in_chunk_order,
                     ordered_sum, shared_fragment_store)
from ..writer import BGZF_EOF, bgzip_file, tabix_index, write_bed_rows, write_bedgraph
from .Occupancy import FragmentMixDistribution, OccupancyParameters, occ_batch

LAST_TIMINGS = {}      # phase -> seconds of the last run_occ call of this process (bench.py's cli_end_to_end reports them)


class _Phases(object):
    """wall-clock seconds per named phase (cheap: two perf_counter calls per phase)"""

    def __init__(self, store):
        import time
        self.store, self.clock = store, time.perf_counter
        store.clear()
        self.t = self.clock()

    def mark(self, name):
        now = self.clock()
        self.store[name] = round(self.store.get(name, 0.0) + now - self.t, 3)
        self.t = now


BATCH_CHUNKS = int(os.environ.get("NATAC_BATCH_CHUNKS", "4096"))   # chunks per sub-batch (the reference maps cores*5 chunks per round)
N_CONTEXTS = int(os.environ.get("NATAC_CONTEXTS", "3"))            # contexts (streams) of the pipelined executor
COMPRESS_LEVEL = 4     # BGZF deflate level of the track files written by the host writer
# Track.write_track + bgzip on the GPU (natac_batch_format_track); NATAC_DEVICE_WRITER=0: the native host writer formats the tracks
DEVICE_WRITER = os.environ.get("NATAC_DEVICE_WRITER", "1") != "0"


def _occHelper(arg):
    """(nuc_dist, OccupancyTrack, [OccPeak]) for one chunk -- same return shape as the reference's helper
    (run_occ.py:23-39); `_occHelperBatch` is what the API offers for lists"""
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


class _Writer(threading.Thread):
    """consumes finished sub-batches in order on its own thread (run_occ.py:41-59 are the reference's writer processes): every
    track is appended to its file -- finished BGZF members when Track.write_track + bgzip ran on the device, else through the
    native host writer (pyatac/tracks.py:37-74) --, then `extra(result)`, then the result's pinned buffers go back to the executor.
    For device-written tracks the tabix records of every result are logged with the byte offset they were written at, so that the
    .tbi can be built without reading the file again (finish_indexes)."""

    def __init__(self, paths, track_of, extra, n_batches, last_rank):
        """extra: a function of the result, or a pair (start, finish): start(result) -> state runs next to the result's file appends
        and before its buffers are released, finish(state) for sub-batch k only after start of sub-batch k + 1 (still in result
        order) -- whatever start handed to a worker pool has company before the writer waits for it"""
        threading.Thread.__init__(self, daemon=True)
        self.paths, self.track_of, self.extra, self.nb, self.last_rank = paths, track_of, extra, n_batches, last_rank
        self.two_phase = isinstance(extra, tuple)
        self._pending = None
        self.q = queue.Queue(maxsize=2)
        self.err = None
        self.seconds = 0.0
        self.seconds_files = 0.0
        self.offset = {n: 0 for n in paths}          # bytes of members written so far (without the EOF marker)
        self.index_log = {n: [] for n in paths}      # (tabix records of a result, offset it was written at)
        self.index_ok = {n: True for n in paths}
        from concurrent.futures import ThreadPoolExecutor
        self.pool = ThreadPoolExecutor(max(1, len(paths)) + 1, thread_name_prefix="natac-track-file")

    def run(self):
        import time
        while True:
            r = self.q.get()
            if r is None:
                return
            try:
                if self.err is None:
                    t0 = time.perf_counter()
                    part = r.tag
                    chroms, starts = [c.chrom for c in part], [c.start for c in part]

                    def write_one(name):
                        path, t = self.paths[name], self.track_of[name]
                        z = r.text.get(t) if r.text else None
                        last = r.seq == self.nb - 1 and self.last_rank
                        if z is not None:       # finished BGZF members from the device: append them (+ the EOF marker at the very end)
                            with open(path, "ab" if r.seq > 0 else "wb") as fh:
                                fh.write(memoryview(z))
                                if last:
                                    fh.write(BGZF_EOF)
                            self.index_log[name].append((r.text_index[t], self.offset[name]))
                            self.offset[name] += len(z)
                        else:
                            write_bedgraph(path, chroms, starts, r.packed.out_off, r.tracks[t], append=r.seq > 0,
                                           compress=COMPRESS_LEVEL, finish=last)
                            self.index_ok[name] = False

                    # one file per track: the appends run side by side (write() releases the GIL), every file still in order; the
                    # per-result extra work (peak rows / calls of THIS result, in result order) runs next to them
                    jobs = [self.pool.submit(write_one, n) for n in self.paths]
                    more = self.pool.submit(self.extra[0] if self.two_phase else self.extra, r)
                    try:
                        for j in jobs:
                            j.result()
                        self.seconds_files += time.perf_counter() - t0
                    finally:                 # the result's buffers are released below: nothing may still be reading them
                        from concurrent.futures import wait
                        wait(jobs + [more])
                    state = more.result()
                    if self.two_phase:
                        r.release()          # start() has copied what finish() needs: the slot goes back before the wait
                        prev, self._pending = self._pending, state
                        if prev is not None:
                            self.extra[1](prev)
                    self.seconds += time.perf_counter() - t0
            except BaseException as e:      # noqa: BLE001 -- re-raised on the main thread
                self.err = e
            finally:
                r.release()

    def put(self, r):
        if self.err is not None:
            raise self.err
        self.q.put(r)

    def finish(self):
        self.q.put(None)
        self.join()
        self.pool.shutdown()
        if self.err is None and self._pending is not None:      # the last sub-batch's second half
            import time
            t0 = time.perf_counter()
            prev, self._pending = self._pending, None
            self.extra[1](prev)
            self.seconds += time.perf_counter() - t0
        if self.err is not None:
            raise self.err


def finish_indexes(writer, names, base_of):
    """.tbi of every track file on rank 0.  `writer`: this rank's _Writer (None without sub-batches).  Files assembled from
    device-formatted members get their index from the logged tabix records -- gathered from all ranks, shifted by the sizes of
    the part files in front --, without being read again; files the host writer touched are indexed by natac_tabix_index."""
    from ..writer import TbiBuilder
    log = dict(index_log=writer.index_log if writer else {n: [] for n in names}, ok=writer.index_ok if writer else {n: True for n in names},
               size=writer.offset if writer else {n: 0 for n in names})
    logs = gather_in_chunk_order([log], dst=0)
    pending = []                # files that need the file-based indexer once rank 0 has assembled them
    if logs is None:
        return pending
    for n in names:
        path = base_of(n)
        if all(l["ok"][n] for l in logs) and DEVICE_WRITER and any(l["index_log"][n] for l in logs):
            tb = TbiBuilder()
            base = 0
            for l in logs:                      # rank order == file order
                for idx, off in l["index_log"][n]:
                    tb.push(idx, base + off)
                base += l["size"][n]
            tb.write(path + ".tbi")
            tb.close()
        else:
            pending.append(path)
    return pending


def run_occ(args):
    ph = _Phases(LAST_TIMINGS)
    if env_rank_world()[2] == 0 and isinstance(args.bam, str):      # the node's publishing rank
        from ..pyatac.fragments import FragmentStore
        FragmentStore.prefetch(args.bam)       # it decodes (shard.shared_fragment_store): start now, next to the FASTA index / BED reads
    if getattr(args, "fasta", None):
        from ..pyatac.seq import FastaStore
        FastaStore.prefetch(args.fasta)        # the genome loads on its own thread; the BED file only needs the record lengths
    chrs = read_chrom_sizes_from_fasta(args.fasta) if args.fasta else read_chrom_sizes_from_bam(args.bam)
    pwm = PWM.open(args.pwm)
    chunks = ChunkList.read(args.bed, chromDict=chrs,
                            min_offset=args.flank + args.upper // 2 + max(pwm.up, pwm.down) + args.nuc_sep // 2)
    chunks.slop(chrs, up=args.nuc_sep // 2, down=args.nuc_sep // 2)
    chunks.merge()
    ensure_distributed()
    rank, world, _ = env_rank_world()
    # global pre-steps once: BAM decode + shared arrays, size histogram (a3) and modelNFR on rank 0
    ph.mark("read_fasta_bed")
    st = shared_fragment_store(args.bam)
    ph.mark("read_bam")
    fragment_dist = None
    if rank == 0:
        fragment_dist = FragmentMixDistribution(0, upper=args.upper)
        if args.sizes is not None:
            tmp = FragmentSizes.open(args.sizes)
            fragment_dist.fragmentsizes = FragmentSizes(0, args.upper, vals=tmp.get(0, args.upper))
        else:
            fragment_dist.getFragmentSizes(st, chunks)
        fragment_dist.modelNFR()
        fragment_dist.fragmentsizes.save(args.out + ".fragmentsizes.txt")
    fragment_dist = broadcast_object(fragment_dist)
    ph.mark("size_hist_modelNFR")
    params = OccupancyParameters(fragment_dist, args.upper, args.fasta, args.pwm, sep=args.nuc_sep, min_occ=args.min_occ,
                                 flank=args.flank, bam=st, ci=args.confidence_interval, step=args.step)
    lens = np.array([c.length() for c in chunks], dtype=np.int64)
    lo, hi = balanced_ranges(lens, np.concatenate(([0], np.cumsum(chunk_fragment_counts(st, chunks)))), world)[rank]
    mine = chunks[lo:hi]
    suffix = "" if world == 1 else ".rank%d" % rank
    track_of = {"occ": L.T_OCC, "occ.lower_bound": L.T_OCC_LOWER, "occ.upper_bound": L.T_OCC_UPPER}
    paths = {n: args.out + "." + n + ".bedgraph.gz" + suffix for n in track_of}
    peaks_path = args.out + ".occpeaks.bed" + suffix
    open(peaks_path, "w").close()
    # sub-batches of <= BATCH_CHUNKS chunks and ~4.5 Mbp (pipeline.sub_batches); an explicit NATAC_BATCH_CHUNKS fixes the chunk count alone
    parts = sub_batches(mine, BATCH_CHUNKS, SUB_BATCH_BP if "NATAC_BATCH_CHUNKS" not in os.environ else 1 << 62)
    dists = []
    if not parts:
        for n in track_of:
            write_bedgraph(paths[n], [], [], [0], np.zeros(0), append=False, compress=COMPRESS_LEVEL, finish=(rank == world - 1))

    def peaks_and_dists(r):
        """OccChunk.callPeaks + getNucDist results of a sub-batch (device: natac_run_occ_peaks): the kept peaks as occpeaks.bed
        rows, the per-chunk nucleosomal size distributions for the ordered sum (run_occ.py:118-123)"""
        part = r.tag
        cc, cp, p_occ, p_lo, p_up, p_rd, keep, nuc_dist = r.occ_peaks
        over = np.nonzero(r.status & 2)[0]
        if len(over):
            # chunks with more local maxima than the device peak finder holds per chunk: the per-chunk API path (host call_peaks)
            from .. import context_lock
            with context_lock:           # writer thread, process-wide context (nucleoatac_amd/__init__.py)
                redo = {int(k): occ_batch([part[int(k)]], params)[0] for k in over}
            fine = ~np.isin(cc, over)
            for k in range(len(part)):           # rows must stay in chunk order: write around the re-done chunks
                if k in redo:
                    with open(peaks_path, "a") as fh:
                        for i in sorted(redo[k].peaks.keys()):
                            redo[k].peaks[i].write(fh)
                    nuc_dist[k] = redo[k].getNucDist()
                else:
                    m = fine & (cc == k) & (keep != 0)
                    _rows(part, cc[m], cp[m], p_occ[m], p_lo[m], p_up[m], p_rd[m])
        else:
            m = keep != 0
            _rows(part, cc[m], cp[m], p_occ[m], p_lo[m], p_up[m], p_rd[m])
        dists.extend(nuc_dist)

    def _rows(part, cc, cp, p_occ, p_lo, p_up, p_rd):
        if not len(cc):
            return
        names = sorted(set(c.chrom for c in part))
        idx = {c: i for i, c in enumerate(names)}
        cid = np.array([idx[c.chrom] for c in part], dtype=np.int32)[cc]
        pos = np.array([c.start for c in part], dtype=np.int64)[cc] + cp
        write_bed_rows(peaks_path, names, cid, pos, pos + 1, np.stack([p_occ, p_lo, p_up, p_rd], axis=1))

    # inside `nucleoatac run` the three tracks also stay in HBM, as the files show them, for the nuc and nfr steps of this process
    resident = None
    if getattr(args, "keep_resident", False) and parts:
        from .. import occstore
        if occstore.ENABLED:
            resident = occstore.OccTrackStore()
            occstore.register(args.out + ".occ.bedgraph.gz", resident)
    if parts:
        stages = Stages(nuc_sd=None, occ=True, ins=None, occ_peaks=dict(min_occ=params.min_occ, sep=params.sep),
                        tracks=() if DEVICE_WRITER else tuple(track_of.values()),
                        text_tracks=tuple(track_of.values()) if DEVICE_WRITER else (),
                        keep=(resident.dev, occstore.TRACKS) if resident is not None else None)
        writer = _Writer(paths, track_of, peaks_and_dists, len(parts), rank == world - 1)
        writer.start()

        pack_s = [0.0]
        arrivals = []

        def pack_part(part):
            import time
            t0 = time.perf_counter()
            pk = pack(part, st, params.fasta, params.chrs, params.pwm if params.fasta is not None else None,
                      window=params.window, upper=params.upper, bias_on_device=True)
            pack_s[0] += time.perf_counter() - t0
            return pk, part

        def items():       # sub-batches packed up to three ahead of the GPU, on their own threads
            return prefetch_map(pack_part, parts, depth=3)

        from .. import default_device
        device = default_device()
        try:
            with PipelinedExecutor(device, lambda ctx: params.occ_calc_params.install(ctx, step=params.step, flank=params.flank),
                                   stages, n_contexts=min(N_CONTEXTS, len(parts))) as ex:
                import time
                t_start, arrivals = time.perf_counter(), []
                for r in ex.map(items()):
                    arrivals.append(time.perf_counter() - t_start)
                    if (r.status & 1).any():
                        k = int(np.flatnonzero(r.status & 1)[0])
                        print("Caught exception when processing:\n" + r.tag[k].asBed() + "\n")
                        r.release()
                        raise ValueError("min() arg is an empty sequence (occupancy likelihood undefined in %s)" % r.tag[k].asBed())
                    if resident is not None:
                        resident.add(r.tag, r.packed.out_off, r.store_seg)
                    writer.put(r)
        finally:
            writer.finish()
        ph.mark("pipeline_wall")
        if arrivals:       # when the results of the sub-batches reached the writer (seconds after the executor started)
            LAST_TIMINGS["results_first_median_gap_last"] = [round(arrivals[0], 3), round(float(np.median(np.diff(arrivals))) if len(arrivals) > 1 else 0.0, 4),
                                                             round(arrivals[-1], 3)]
        LAST_TIMINGS["pack_inside_pipeline"] = round(pack_s[0], 3)
        LAST_TIMINGS["writer_inside_pipeline"] = round(writer.seconds, 3)
        LAST_TIMINGS["file_appends_inside_writer"] = round(writer.seconds_files, 3)
    dists = gather_in_chunk_order(dists, dst=0)
    peaks_job, peaks_err = None, []
    if world == 1:      # one rank: occpeaks.bed is complete -- its bgzip + tabix (run_occ.py:130-136) run next to the track indexes
        def _peaks():
            try:
                bgzip_file(args.out + ".occpeaks.bed", level=COMPRESS_LEVEL)
                tabix_index(args.out + ".occpeaks.bed.gz")
            except BaseException as e:      # noqa: BLE001 -- re-raised below, on the main thread
                peaks_err.append(e)
        peaks_job = threading.Thread(target=_peaks, name="natac-occpeaks", daemon=True)
        peaks_job.start()
    to_index = finish_indexes(writer if parts else None, list(track_of), lambda n: args.out + "." + n + ".bedgraph.gz")
    ph.mark("gather_and_track_indexes")
    barrier()      # every rank has closed its part files (raises if WORLD_SIZE > 1 without a process group)
    if rank == 0:
        if world > 1:   # BGZF members / text lines concatenate: rank order == chunk order
            for n in list(track_of) + ["occpeaks"]:
                base = args.out + "." + n + (".bed" if n == "occpeaks" else ".bedgraph.gz")
                with open(base, "wb") as fo:
                    for r in range(world):
                        with open(base + ".rank%d" % r, "rb") as fi:
                            shutil.copyfileobj(fi, fo)
                        os.remove(base + ".rank%d" % r)
        # bgzip + tabix of every output like the reference (run_occ.py:130-136)
        ph.mark("merge_part_files")
        if peaks_job is not None:
            peaks_job.join()
            if peaks_err:
                raise peaks_err[0]
        else:
            bgzip_file(args.out + ".occpeaks.bed", level=COMPRESS_LEVEL)
            tabix_index(args.out + ".occpeaks.bed.gz")
        ph.mark("occpeaks_bgzip_tabix")
        for path in to_index:
            tabix_index(path)
        nuc_dist = ordered_sum(dists) if dists else np.zeros(args.upper)
        FragmentSizes(0, args.upper, vals=nuc_dist).save(args.out + ".nuc_dist.txt")
    ph.mark("nuc_dist_and_rest")
