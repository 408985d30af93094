"""Executors above the C-ABI: how sub-batches of packed chunks move through one GPU.

Two shapes, both product code (bench.py and the CLI drivers call these; nothing here is benchmark-only):

  ResidentShard      every sub-batch of a rank's shard is uploaded once and stays in HBM; `step()` runs the stages over all
                     of them.  When the outputs of all sub-batches do not fit next to the inputs (BASELINE configs[3] on
                     one GPU: 3 Gbp x ~175 B per base) the outputs of a sub-batch are handed to `consume` and then released
                     (`natac_batch_release_outputs`), so the next sub-batch's stages reuse the same pool blocks.

  PipelinedExecutor  host -> device -> host streaming: N host threads, each with its own natac context (= HIP stream) on the
                     same GPU and its own page-locked output slots, take sub-batches in turn -- upload, stages, download --
                     so that uploads, kernels and downloads of different sub-batches overlap on the device.  Results come
                     back IN INPUT ORDER; the consumer (a track writer) runs on the caller's thread or its own, so writing
                     overlaps with compute the way the reference's writer processes do behind their JoinableQueues
                     (nucleoatac/run_occ.py:109-123, run_nuc.py:170-188).
"""
import collections
import threading

import numpy as np

from . import _lib as L
from .device import Context, pinned_empty

# bytes of device outputs per base when every stage has run: 11 f64 per-base tracks + the background kernel's two internal
# window sums + int32 insertions + grid / block arrays of the occupancy stage (~175 B)
OUT_BYTES_PER_BP = 175.0


class Stages(object):
    """which stages run on a sub-batch and what is brought back to the host.

    nuc_sd      NucParameters.smooth_sd for natac_run_nuc, or None to skip the nuc stage
    occ         run natac_run_occ
    ins         (lower, upper) for natac_run_ins, or None
    peaks       kwargs of DeviceBatch.run_peaks (candidate search + LR / var / z), or None
    occ_peaks   kwargs of DeviceBatch.run_occ_peaks (OccChunk.callPeaks + getNucDist), or None
    tracks      per-base tracks to download as arrays (NATAC_T_* ids)
    keep        (TrackStore, tracks): a copy of these tracks, as their .bedgraph file shows them, stays in HBM (occstore.py)
    text_tracks per-base tracks to bring back as finished bedGraph.gz bytes instead: Track.write_track + bgzip run on the device
                (natac_batch_format_track); the sub-batch needs `chroms` and `chunk_start` (pipeline.pack sets them)
    """

    def __init__(self, nuc_sd=10, occ=True, ins=(0, 2000), peaks=None, occ_peaks=None, tracks=(), text_tracks=(), keep=None):
        self.nuc_sd, self.occ, self.ins, self.peaks, self.occ_peaks = nuc_sd, occ, ins, peaks, occ_peaks
        self.keep = keep      # (device.TrackStore, track ids): adopt these tracks into the store (Result.store_seg)
        self.tracks = tuple(int(t) for t in tracks)
        self.text_tracks = tuple(int(t) for t in text_tracks)

    def run(self, batch):
        """launch every stage on `batch` (asynchronous except the candidate counts); returns the number of nuc candidates
        (0 without a peaks stage).  Candidate arrays stay in HBM (download_peaks)."""
        if self.nuc_sd is not None:
            batch.run_nuc(self.nuc_sd)
        if self.occ:
            batch.run_occ()
        if self.ins is not None:
            batch.run_ins(*self.ins)
        n = 0
        if self.peaks is not None:
            n = batch.run_peaks(download=False, **self.peaks)
        return n


class ResidentShard(object):
    """the sub-batches of one rank's shard, inputs resident in HBM.

    recycle: "auto" = release a sub-batch's outputs after its stages when all outputs would not fit in `mem_fraction` of the
    device memory next to the inputs; True / False force it.  With recycling, `consume(i, batch, n_cand)` (if given) is called
    before the release -- the only moment the outputs of sub-batch i exist."""

    def __init__(self, ctx, subs, recycle="auto", mem_fraction=0.8):
        self.ctx = ctx
        self.batches = [ctx.upload(s) for s in subs]
        ctx.sync()
        self.total_bp = sum(s.total_bp for s in subs)
        if recycle == "auto":
            mem = ctx.device_info()["mem_bytes"]
            recycle = len(self.batches) > 1 and self.total_bp * OUT_BYTES_PER_BP > mem_fraction * mem
        self.recycle = bool(recycle)
        self.last_n = [0] * len(self.batches)

    def step(self, stages, consume=None):
        """one pass of `stages` over every sub-batch; returns the total number of candidates"""
        total = 0
        for i, b in enumerate(self.batches):
            n = stages.run(b)
            self.last_n[i] = n
            total += n
            if consume is not None:
                consume(i, b, n)
            if self.recycle:
                b.release_outputs()
        return total

    def close(self):
        for b in self.batches:
            b.free()
        self.batches = []


class Result(object):
    """outputs of one sub-batch in page-locked host memory; `release()` hands the buffers back to the executor"""
    __slots__ = ("seq", "packed", "tag", "tracks", "text", "text_index", "peaks", "occ_peaks", "status", "store_seg", "_slot", "_ex")

    def release(self):
        if self._slot is not None:
            self._ex._free_slot(self._slot)
            self._slot = None
        self.tracks = self.text = self.text_index = self.peaks = self.occ_peaks = None


class _Slot(object):
    def __init__(self, owner):
        self.owner = owner
        self.bufs = {}

    def view(self, t, n, dtype):
        """a length-n pinned array for track t (grown on demand; steady state allocates nothing)"""
        cur = self.bufs.get(t)
        if cur is None or cur.size < n or cur.dtype != np.dtype(dtype):
            cur = self.bufs[t] = pinned_empty(int(n * 1.05) + 16, dtype)
        return cur[:n]


class PipelinedExecutor(object):
    """N contexts on one GPU; `map(iterable of (PackedChunks, tag))` yields `Result`s in input order.

    configure(ctx) installs the run constants (VMat, sizes, occupancy model) on a fresh context.  Every worker owns
    `slots_per_context` output slots, so up to n_contexts * slots_per_context results can be alive (downloaded, being
    written) at once; a worker only takes the next sub-batch when it holds a free slot, which keeps the lowest unfinished
    sequence number always in progress (no deadlock with an in-order consumer)."""

    def __init__(self, device, configure, stages, n_contexts=4, slots_per_context=2):
        self.device, self.configure, self.stages = int(device), configure, stages
        self.n_contexts, self.slots_per_context = int(n_contexts), int(slots_per_context)
        self._lock = threading.Condition()
        self._free = collections.defaultdict(list)       # worker id -> free slots
        self._done = {}
        self._err = None
        self._ctxs = {}                                  # worker id -> Context (kept across map() calls until close())
        self.bytes_down = self.bytes_up = 0

    def close(self):
        """destroy the contexts and drop the pinned slots"""
        for c in self._ctxs.values():
            c.close()
        self._ctxs = {}
        self._free.clear()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- slots -------------------------------------------------------------------------------------
    def _free_slot(self, slot):
        with self._lock:
            self._free[slot.owner].append(slot)
            self._lock.notify_all()

    def _take_slot(self, wid):
        with self._lock:
            while not self._free[wid] and self._err is None:
                self._lock.wait()
            if self._err is not None:
                raise RuntimeError("pipeline aborted") from self._err
            return self._free[wid].pop()

    # ---- one sub-batch -------------------------------------------------------------------------------
    def _process(self, ctx, slot, seq, packed, tag):
        st = self.stages
        b = ctx.upload(packed)
        try:
            n = st.run(b)
            r = Result()
            r.seq, r.packed, r.tag, r._slot, r._ex = seq, packed, tag, slot, self
            r.tracks = {}
            for t in st.tracks:
                dt = np.int32 if t == L.T_INS else np.float64
                r.tracks[t] = b.track(t, out=slot.view(t, b.total_bp, dt))
            r.text, r.text_index = {}, {}
            for t in st.text_tracks:
                # Track.write_track + bgzip on the device: BGZF members straight into the pinned slot
                # (the copy of a finished track into the pinned slot runs while the next track is formatted: wait=False + format_wait below)
                buf, info = b.format_track(t, packed.chroms, packed.chunk_start, compress=True,
                                           out=lambda nbytes, t=t: slot.view(("z", t), nbytes, np.uint8), wait=False)
                if info["hard"]:        # a value whose 12th digit the device table cannot decide (|v| >= 1e12 ties): host formatter
                    if t not in r.tracks:
                        dt = np.int32 if t == L.T_INS else np.float64
                        r.tracks[t] = b.track(t, out=slot.view(t, b.total_bp, dt))
                    r.text[t] = None
                else:
                    r.text[t] = buf
                    r.text_index[t] = info["index"]      # tabix records of these members (writer.TbiBuilder)
            if st.text_tracks:
                b.format_wait()
            r.peaks = b.download_peaks(n) if st.peaks is not None else None
            r.occ_peaks = b.run_occ_peaks(**st.occ_peaks) if st.occ_peaks is not None else None
            r.status = b.status()
            r.store_seg = st.keep[0].adopt(b, st.keep[1]) if st.keep is not None else None
            down = sum(a.nbytes for a in r.tracks.values()) + (n * 32 if st.peaks is not None else 0) + \
                sum(a.nbytes for a in r.text.values() if a is not None)
            up = packed.frag_lpos.nbytes + packed.frag_ilen.nbytes + (packed.bias_log.nbytes if packed.bias_log is not None else 0)
            with self._lock:
                self.bytes_down += down
                self.bytes_up += up
            return r
        finally:
            b.free()

    def _worker(self, wid, source):
        ctx = None
        try:
            ctx = self._ctxs.get(wid)
            if ctx is None:
                ctx = Context(self.device)
                self.configure(ctx)
                with self._lock:
                    self._ctxs[wid] = ctx
                    for _ in range(self.slots_per_context):
                        self._free[wid].append(_Slot(wid))
            while True:
                slot = self._take_slot(wid)
                item = source()
                if item is None:
                    self._free_slot(slot)
                    break
                seq, packed, tag = item
                r = self._process(ctx, slot, seq, packed, tag)
                with self._lock:
                    self._done[seq] = r
                    self._lock.notify_all()
        except BaseException as e:      # noqa: BLE001 -- handed to the consumer thread
            with self._lock:
                if self._err is None:
                    self._err = e
                self._lock.notify_all()
        finally:
            if ctx is not None:
                try:
                    ctx.sync()
                except Exception:       # noqa: BLE001
                    pass

    def map(self, items):
        """generator of Results in the order of `items` (an iterable of (PackedChunks, tag)).  The iterable is advanced by the
        worker threads, one item at a time under a lock, so host-side packing of the next sub-batch overlaps with the GPU
        work of the previous ones.  The consumer must `release()` every Result (after writing it)."""
        it = iter(items)
        with self._lock:
            stale, self._done, self._err = list(self._done.values()), {}, None
        for r in stale:                  # results of an abandoned map(): their slots go back
            r.release()
        src_lock = threading.Lock()
        state = {"next": 0, "end": None}

        def source():
            with src_lock:
                if state["end"] is not None:
                    return None
                try:
                    packed, tag = next(it)
                except StopIteration:
                    state["end"] = state["next"]
                    with self._lock:
                        self._lock.notify_all()
                    return None
                except BaseException as e:      # noqa: BLE001 -- a packing error ends the run on the consumer thread
                    state["end"] = state["next"]
                    with self._lock:
                        if self._err is None:
                            self._err = e
                        self._lock.notify_all()
                    return None
                seq = state["next"]
                state["next"] += 1
                return seq, packed, tag

        threads = [threading.Thread(target=self._worker, args=(w, source), daemon=True) for w in range(self.n_contexts)]
        for t in threads:
            t.start()
        want = 0
        try:
            while True:
                with self._lock:
                    while want not in self._done and self._err is None and not (state["end"] is not None and want >= state["end"]):
                        self._lock.wait()
                    if self._err is not None:
                        raise self._err
                    if want not in self._done:
                        break
                    r = self._done.pop(want)
                want += 1
                yield r
        finally:
            with self._lock:
                if self._err is None and (state["end"] is None or want < state["end"]):
                    self._err = GeneratorExit("consumer stopped early")
                self._lock.notify_all()
            with src_lock:
                if state["end"] is None:
                    state["end"] = state["next"]
            for t in threads:
                t.join()
            if isinstance(self._err, GeneratorExit):
                self._err = None
