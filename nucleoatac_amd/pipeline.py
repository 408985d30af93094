"""Batched execution of the occ + nuc path: ChunkList + FragmentStore (+ FASTA) -> PackedChunks -> GPU -> per-chunk
results in chunk order.  This is what replaces `pool.map(_occHelper / _nucHelper, ...)` of the reference
(nucleoatac/run_occ.py:118-123, run_nuc.py:183-187): thousands of chunks per kernel launch instead of one
process per chunk."""
import numpy as np

from . import _lib as L
from . import get_context
from .packing import BIAS_LEFT, BIAS_RIGHT, PackedChunks, sort_by_centre
from .pyatac.bias import InsertionBiasTrack
from .pyatac.fragments import FragmentStore

# fragments attached to a chunk: l in [start - MARGIN, end + MARGIN) -- covers every fragment that can put a centre
# within 126 bp of the chunk or an insertion inside it for insert sizes < 2000 (pyatac/fragments.pyx:24, tracks.py:164)
MARGIN = 2000 + 126


def pack(chunks, bam, fasta=None, chrs=None, pwm=None, atac=True, margin=MARGIN):
    """PackedChunks for a list of Chunk objects.  The log-bias slice of every chunk is the PWM score of
    [start-246, end+247) (InsertionBiasTrack.computeBias, as in nucleoatac/Occupancy.py:212-214) or None."""
    st = FragmentStore.open(bam)
    starts, lens, offs, ls, ns, boffs, bvals, chroms = [], [], [0], [], [], [0], [], []
    bias_of = _bias_spans(chunks, fasta, chrs, pwm) if fasta is not None else None
    for ch in chunks:
        l, n = st.fetch(ch.chrom, ch.start - margin, ch.end + margin, 1 if atac else 0)
        keep = l >= ch.start - margin
        l, n = l[keep], n[keep]
        lr = (l - ch.start).astype(np.int32)
        o = sort_by_centre(lr, n)
        ls.append(lr[o])
        ns.append(n[o].astype(np.int32))
        offs.append(offs[-1] + len(lr))
        starts.append(ch.start)
        lens.append(ch.end - ch.start)
        chroms.append(ch.chrom)
        if fasta is not None:
            vals = bias_of(ch)
            bvals.append(vals)
            boffs.append(boffs[-1] + len(vals))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return PackedChunks(chunk_start=np.array(starts, np.int64), chunk_len=np.array(lens, np.int32),
                        frag_off=np.array(offs, np.int64), frag_lpos=cat(ls, np.int32), frag_ilen=cat(ns, np.int32),
                        bias_off=np.array(boffs, np.int64) if fasta is not None else None,
                        bias_log=cat(bvals, np.float64) if fasta is not None else None, chroms=chroms)


def _bias_spans(chunks, fasta, chrs, pwm, max_gap=4096):
    """PWM log-bias for [start-246, end+247) of every chunk (InsertionBiasTrack.computeBias, pyatac/bias.py:85-92).
    Windows of nearby chunks are merged into spans that are scored with ONE natac_pwm_bias launch each and then sliced,
    instead of one sequence fetch + launch per chunk as in the reference (Occupancy.py:212-214)."""
    by_chrom = {}
    for ch in chunks:
        by_chrom.setdefault(ch.chrom, []).append((ch.start - BIAS_LEFT, ch.end + BIAS_RIGHT))
    spans = {}
    for chrom, iv in by_chrom.items():
        iv.sort()
        cur = list(iv[0])
        merged = []
        for a, b in iv[1:]:
            if a <= cur[1] + max_gap:
                cur[1] = max(cur[1], b)
            else:
                merged.append(cur)
                cur = [a, b]
        merged.append(cur)
        tracks = []
        for a, b in merged:
            bt = InsertionBiasTrack(chrom, a, b, log=True)
            bt.computeBias(fasta, chrs, pwm)
            tracks.append((bt.start, bt.end, bt.vals))
        spans[chrom] = (np.array([t[0] for t in tracks]), tracks)

    def lookup(ch):
        a, b = ch.start - BIAS_LEFT, ch.end + BIAS_RIGHT
        s0, tracks = spans[ch.chrom]
        t = tracks[int(np.searchsorted(s0, a, "right")) - 1]
        if a < t[0] or b > t[1]:
            raise Exception("chunk %s too close to the chromosome end for the bias window" % ch.asBed())
        return t[2][a - t[0]:b - t[0]]

    return lookup


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
            raise ValueError("min() arg is an empty sequence (occupancy likelihood undefined in chunk %d)" % k)
        out = {}
        self.flat = getattr(self, "flat", {})
        for name, t in (("smoothed_vals", L.T_OCC), ("smoothed_lower", L.T_OCC_LOWER), ("smoothed_upper", L.T_OCC_UPPER),
                        ("cov", L.T_OCC_COV), ("smoothed_prefill", L.T_OCC_PREFILL)):
            self.flat[name] = b.track(t)
            out[name] = b.split(self.flat[name])
        grids = [b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)]
        step = self.ctx.occ_step
        half = (step - 1) // 2
        off = 0
        out["vals"], out["lower_bound"], out["upper_bound"] = [], [], []
        for k in range(self.pk.n_chunks):
            Lk = int(self.pk.chunk_len[k])
            nk = len(range(half, Lk, step))
            for key, g in zip(("vals", "lower_bound", "upper_bound"), grids):
                v = np.full(Lk, np.nan)
                v[:min(Lk, nk * step)] = np.repeat(g[off:off + nk], step)[:min(Lk, nk * step)]
                out[key].append(v)
            off += nk
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
