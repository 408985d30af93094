"""Batched execution of the occ + nuc path: ChunkList + FragmentStore (+ FASTA) -> PackedChunks -> GPU -> per-chunk
results in chunk order.  This is what replaces `pool.map(_occHelper / _nucHelper, ...)` of the reference
(nucleoatac/run_occ.py:118-123, run_nuc.py:183-187): thousands of chunks per kernel launch instead of one
process per chunk."""
import numpy as np

from . import _lib as L
from . import get_context
from .packing import BIAS_LEFT, BIAS_RIGHT, PackedChunks, sort_by_centre
from .pyatac.fragments import FragmentStore

# fragments attached to a chunk: l in [start - MARGIN, end + MARGIN) -- covers every fragment that can put a centre
# within 126 bp of the chunk or an insertion inside it for insert sizes < 2000 (pyatac/fragments.pyx:24, tracks.py:164)
MARGIN = 2000 + 126


def prefetch_map(fn, items, depth=3):
    """fn(item) for every item, in order, computed up to `depth` items ahead on a small thread pool: host-side packing of the
    next sub-batches (numpy + the native packer, both release the GIL) overlaps with each other and with the consumer"""
    import collections
    import itertools
    from concurrent.futures import ThreadPoolExecutor
    it = iter(items)
    with ThreadPoolExecutor(max(1, depth), thread_name_prefix="natac-pack") as ex:
        futs = collections.deque(ex.submit(fn, x) for x in itertools.islice(it, max(1, depth)))
        while futs:
            res = futs.popleft().result()
            nxt = next(it, None)
            if nxt is not None:
                futs.append(ex.submit(fn, nxt))
            yield res


def bias_window(window, upper):
    """(left, right) extent of the reference's bias track around a chunk: [start - window - upper//2, end + window +
    upper//2 + 1) with window = 2*flank+1 (occ, Occupancy.py:184, 212-213) or the V-plot width (nuc,
    NucleosomeCalling.py:214, 246-247); never below the packing default 246 / 247 so that occ and nuc batches of the
    default parameters share one layout."""
    ext = int(window) + int(upper) // 2
    return max(BIAS_LEFT, ext), max(BIAS_RIGHT, ext + 1)


def chunk_fragment_counts(st, chunks):
    """number of reads near every chunk (pos in [start - 1024, end), what FragmentStore.fetch returns): the fragment term of
    the shard balance.  One searchsorted per chromosome instead of one fetch per chunk."""
    st = FragmentStore.open(st)
    n = np.zeros(len(chunks), dtype=np.int64)
    chroms = np.array([c.chrom for c in chunks])
    starts = np.array([c.start for c in chunks], dtype=np.int64)
    ends = np.array([c.end for c in chunks], dtype=np.int64)
    for chrom in set(chroms.tolist()):
        if chrom not in st.pos:
            continue
        m = np.nonzero(chroms == chrom)[0]
        p = st.pos[chrom]
        n[m] = np.searchsorted(p, ends[m], "left") - np.searchsorted(p, starts[m] - 1024, "left")
    return n


def pack(chunks, bam, fasta=None, chrs=None, pwm=None, atac=True, margin=MARGIN, window=None, upper=None, bias_on_device=False):
    """PackedChunks for a list of Chunk objects.  The log-bias slice of every chunk is the PWM score of
    [start-246, end+247) (InsertionBiasTrack.computeBias, as in nucleoatac/Occupancy.py:212-214) or None; `window` /
    `upper` (OccupancyParameters.window / NucParameters.window and .upper) widen it for non-default parameters."""
    bl, br = bias_window(window, upper) if window is not None and upper is not None else (BIAS_LEFT, BIAS_RIGHT)
    st = FragmentStore.open(bam)
    nc = len(chunks)
    chroms = [ch.chrom for ch in chunks]
    starts = np.array([ch.start for ch in chunks], np.int64)
    ends = np.array([ch.end for ch in chunks], np.int64)
    offs, lpos, ilen = _pack_fragments(st, chroms, starts, ends, margin, atac)
    boffs = bias = None
    extra = {}
    if fasta is not None and bias_on_device:
        # hand the sequence windows to the device: the PWM score lands in the batch's bias array without a round trip
        seq_off, seq = _seq_windows(chunks, starts, ends, fasta, pwm, bl, br)
        lens = set(len(x) for x in pwm.nucleotides)
        if lens != {1}:
            raise NotImplementedError("k-mer PWMs are not supported on the device (single-nucleotide PWMs only)")
        extra = dict(seq_off=seq_off, seq=seq, pwm_log=np.log(np.asarray(pwm.mat, dtype=np.float64)),
                     pwm_nucs=np.frombuffer("".join(pwm.nucleotides).encode("ascii"), dtype=np.uint8))
    elif fasta is not None:
        boffs, bias = _bias_batch(chunks, starts, ends, fasta, pwm, bl, br)
    return PackedChunks(chunk_start=starts, chunk_len=(ends - starts).astype(np.int32), frag_off=offs, frag_lpos=lpos,
                        frag_ilen=ilen, bias_off=boffs, bias_log=bias, chroms=chroms, bias_left=bl, bias_right=br, **extra)


def _pack_fragments(st, chroms, starts, ends, margin, atac):
    """CSR fragment arrays of the chunks through natac_pack_chunks (count pass + multi-threaded fill)"""
    import ctypes as C
    lib = L.load()
    nc = len(chroms)
    refs = st.references
    idx = {c: i for i, c in enumerate(refs)}
    cid = np.array([idx.get(c, -1) for c in chroms], np.int32)
    nref = len(refs)
    pos_p = (C.c_void_p * max(1, nref))(*[st.pos[c].ctypes.data for c in refs])
    tl_p = (C.c_void_p * max(1, nref))(*[st.tlen[c].ctypes.data for c in refs])
    npc = np.array([len(st.pos[c]) for c in refs], np.int64)
    offs = np.zeros(nc + 1, np.int64)
    first = np.zeros(max(1, nc), np.int64)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    args = (nc, vp(starts), vp(ends), vp(cid), nref, C.cast(pos_p, C.c_void_p), C.cast(tl_p, C.c_void_p), vp(npc), int(margin),
            1 if atac else 0, vp(offs), vp(first))
    L.check(lib.natac_pack_chunks(*args, None, None, 0))
    lpos = np.empty(int(offs[-1]), np.int32)
    ilen = np.empty(int(offs[-1]), np.int32)
    L.check(lib.natac_pack_chunks(*args, vp(lpos), vp(ilen), 0))
    return offs, lpos, ilen


def _seq_windows(chunks, starts, ends, fasta, pwm, bias_left, bias_right):
    """(seq_off, seq): the sequence windows [start - bias_left - pwm.up, end + bias_right + pwm.down) of the chunks laid end to end"""
    from .pyatac.seq import FastaStore
    fs = FastaStore.open(fasta)
    a = starts - bias_left - pwm.up
    b = ends + bias_right + pwm.down
    segs = []
    for k, ch in enumerate(chunks):
        s = fs.seqs.get(ch.chrom)
        if s is None or a[k] < 0 or b[k] > len(s):
            raise Exception("chunk %s too close to the chromosome end for the bias window" % ch.asBed())
        segs.append(s[int(a[k]):int(b[k])])
    off = np.zeros(len(chunks) + 1, np.int64)
    np.cumsum(b - a, out=off[1:])
    return off, (np.concatenate(segs) if segs else np.zeros(0, np.uint8))


def _bias_batch(chunks, starts, ends, fasta, pwm, bias_left, bias_right):
    """PWM log-bias of [start - bias_left, end + bias_right) for every chunk of a sub-batch with ONE natac_pwm_bias launch
    (InsertionBiasTrack.computeBias, pyatac/bias.py:85-92, per chunk in the reference, Occupancy.py:212-214): the sequence windows
    [start - bias_left - pwm.up, end + bias_right + pwm.down) of all chunks are laid end to end, scored in one pass, and the
    K - 1 scores that straddle two windows are dropped by the gather that forms the packed bias array."""
    from . import get_context
    K = pwm.up + pwm.down + 1
    _, cat = _seq_windows(chunks, starts, ends, fasta, pwm, bias_left, bias_right)
    scores = get_context().pwm_bias(cat, pwm.mat, pwm.nucleotides)
    lens = (ends - starts) + bias_left + bias_right                  # scores kept per chunk
    boffs = np.zeros(len(chunks) + 1, np.int64)
    np.cumsum(lens, out=boffs[1:])
    idx = np.arange(int(boffs[-1]), dtype=np.int64) + np.repeat(np.arange(len(chunks), dtype=np.int64) * (K - 1), lens)
    return boffs, scores[idx]


def window_size_hist(pk, k, pos, flank, upper):
    """counts by insert size of the fragments of chunk k centred within +-flank of `pos` (relative): the row sums of
    FragmentMat2D.get(start=pos-flank, end=pos+flank+1) used by OccChunk.getNucDist (nucleoatac/Occupancy.py:232-240)"""
    l, n = pk.chunk_frags(k)
    c = l.astype(np.int64) + (n.astype(np.int64) - 1) // 2
    a, b = np.searchsorted(c, pos - flank, "left"), np.searchsorted(c, pos + flank, "right")
    nn = n[a:b]
    nn = nn[(nn >= 0) & (nn < upper)]
    return np.bincount(nn, minlength=upper).astype(np.float64)


class BatchRunner(object):
    """uploads one PackedChunks batch and exposes the per-chunk outputs of the GPU stages"""

    def __init__(self, packed, ctx=None):
        self.ctx = ctx or get_context()
        self.pk = packed
        self.batch = self.ctx.upload(packed)

    def close(self):
        self.batch.free()

    def occ(self):
        b = self.batch
        b.run_occ()
        st = b.status()
        if st.any():
            k = int(np.flatnonzero(st)[0])
            b.free()
            where = "%s\t%d\t%d" % (self.pk.chroms[k], int(self.pk.chunk_start[k]), int(self.pk.chunk_start[k]) + int(self.pk.chunk_len[k])) \
                if self.pk.chroms else "chunk %d of the batch" % k
            # the reference's helper prints the chunk's BED line before re-raising (run_occ.py:33-37)
            raise ValueError("min() arg is an empty sequence (occupancy likelihood undefined in %s)" % where)
        out = {}
        self.flat = getattr(self, "flat", {})
        for name, t in (("smoothed_vals", L.T_OCC), ("smoothed_lower", L.T_OCC_LOWER), ("smoothed_upper", L.T_OCC_UPPER),
                        ("cov", L.T_OCC_COV)):
            self.flat[name] = b.track(t)
            out[name] = b.split(self.flat[name])
        grids = [b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)]
        # grid point k of a chunk covers bases [k*step, min((k+1)*step, L)) (Occupancy.py:136-146); bases past the last grid
        # point's block stay NaN.  One gather for the whole batch instead of an np.repeat per chunk.
        step = self.ctx.occ_step
        half = (step - 1) // 2
        lens = self.pk.chunk_len.astype(np.int64)
        nk = np.where(lens > half, (lens - half + step - 1) // step, 0)
        goff = np.concatenate(([0], np.cumsum(nk)))
        rel = np.arange(self.pk.total_bp, dtype=np.int64) - np.repeat(self.pk.out_off[:-1], lens)
        k = rel // step
        ok = k < np.repeat(nk, lens)
        gi = np.where(ok, np.repeat(goff[:-1], lens) + k, 0)
        for key, g in zip(("vals", "lower_bound", "upper_bound"), grids):
            flat = np.where(ok, g[gi] if len(g) else np.nan, np.nan)
            out[key] = b.split(flat)
        return out

    def nuc(self, smooth_sd):
        b = self.batch
        b.run_nuc(smooth_sd)
        self.flat = getattr(self, "flat", {})
        out = {}
        for name, t in (("nuc_cov", L.T_NUC_COV), ("nfr_cov", L.T_NFR_COV), ("nuc_signal", L.T_RAW), ("bias", L.T_BACKGROUND),
                        ("norm_signal", L.T_NORM), ("smoothed", L.T_SMOOTH)):
            self.flat[name] = b.track(t)
            out[name] = b.split(self.flat[name])
        return out

    def ins(self, lower=0, upper=2000):
        self.batch.run_ins(lower, upper)
        return [x.astype(np.float64) for x in self.batch.split(self.batch.track(L.T_INS))]

    def candidates(self, cand_chunk, cand_pos):
        return self.batch.run_candidates(cand_chunk, cand_pos)


# bases per sub-batch of `occ`'s pipeline: small enough that the device arrays and the pinned output slots of a sub-batch (30 bytes of
# compressed track per base and track) are allocated in a blink, the first result arrives after < 0.1 s and the writer thread is fed
# evenly; large enough that the per-launch and per-call overheads stay small.  tools/occ_batch_sweep.py, three contexts, seconds of
# `nucleoatac occ` (pipeline wall):   60 k x 10 kb tiles:  41 Mbp (4,096 chunks, the old rule) 4.3-4.8 (3.0) | 18 Mbp 3.33 (2.24) |
# 9 Mbp 2.97 (2.06) | 4.5 Mbp 2.49 (1.63) | 3 Mbp 2.50 (1.64) | 2.2 Mbp 2.71 (1.82) | 1.1 Mbp 2.78 (1.92);
# 100 k x 2 kb windows:  18 Mbp 1.83 (1.06) | 9 Mbp (4,096 chunks, the old rule) 1.83 (1.09) | 4.5 Mbp 1.54 (0.83) | 2.2 Mbp 1.57 (0.91).
SUB_BATCH_BP = 4500000


def sub_batches(chunks, max_chunks=4096, target_bp=SUB_BATCH_BP):
    """consecutive slices of `chunks` (a list / ChunkList), each at most `max_chunks` chunks and -- unless a single chunk is longer --
    at most `target_bp` bases: the reference maps `cores * 5` chunks per round whatever their length (run_occ.py:101-123)"""
    out, a, bp = [], 0, 0
    for i, c in enumerate(chunks):
        n = c.end - c.start
        if i > a and (i - a >= max_chunks or bp + n > target_bp):
            out.append(chunks[a:i])
            a, bp = i, 0
        bp += n
    if len(chunks) > a:
        out.append(chunks[a:len(chunks)])
    return out
