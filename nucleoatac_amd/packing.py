"""Packed per-chunk fragment / bias arrays: the host-side layout handed to the C-ABI.

The reference hands every worker a BAM path and lets it re-open and re-scan the file
per chunk (pyatac/fragments.pyx:21-25).  Here the scan happens once; each chunk gets
a contiguous slice of two int32 arrays (SoA) sorted by fragment centre, plus a
contiguous slice of the float64 log-bias track.  All coordinates inside a chunk are
relative to the chunk start so they fit int32 on any genome.

Layout (little-endian, C-contiguous):
  chunk_start int64[nc]      genomic start of chunk (after slop+merge)
  chunk_len   int32[nc]      L = end - start
  frag_off    int64[nc+1]    CSR offsets into frag_lpos / frag_ilen
  frag_lpos   int32[nf]      l - chunk_start   (l = pos+4, pyatac/fragments.pyx:28)
  frag_ilen   int32[nf]      n = |tlen|-8      (pyatac/fragments.pyx:31)
                             sorted by centre c = l + (n-1)//2 within each chunk
  bias_off    int64[nc+1]    CSR offsets into bias_log (or None: no FASTA => bias 1)
  bias_log    float64[nb]    log Tn5 bias for [start - bias_left, end + bias_right)
  out_off     int64[nc+1]    prefix sum of chunk_len (per-base outputs are concatenated)
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# reference bias-track window: [start - window - upper//2, end + window + upper//2 + 1) with
# window = 121, upper = 251 (nucleoatac/NucleosomeCalling.py:246-247, Occupancy.py:212-213)
BIAS_LEFT = 246
BIAS_RIGHT = 247


@dataclass
class PackedChunks:
    chunk_start: np.ndarray
    chunk_len: np.ndarray
    frag_off: np.ndarray
    frag_lpos: np.ndarray
    frag_ilen: np.ndarray
    bias_off: Optional[np.ndarray]
    bias_log: Optional[np.ndarray]
    bias_left: int = BIAS_LEFT
    bias_right: int = BIAS_RIGHT
    chroms: List[str] = field(default_factory=list)
    out_off: np.ndarray = None
    # alternative to bias_log: the genome sequence of every chunk's bias window, scored on the device when the batch is uploaded
    # (natac_batch_create_from_seq): seq[seq_off[i]:seq_off[i+1]] = bases of [start - bias_left - up, end + bias_right + down)
    seq_off: Optional[np.ndarray] = None
    seq: Optional[np.ndarray] = None
    pwm_log: Optional[np.ndarray] = None       # log(PWM.mat), (nrow, K)
    pwm_nucs: Optional[np.ndarray] = None      # uint8 row letters

    def __post_init__(self):
        self.chunk_start = np.ascontiguousarray(self.chunk_start, dtype=np.int64)
        self.chunk_len = np.ascontiguousarray(self.chunk_len, dtype=np.int32)
        self.frag_off = np.ascontiguousarray(self.frag_off, dtype=np.int64)
        self.frag_lpos = np.ascontiguousarray(self.frag_lpos, dtype=np.int32)
        self.frag_ilen = np.ascontiguousarray(self.frag_ilen, dtype=np.int32)
        if self.bias_log is not None:
            self.bias_off = np.ascontiguousarray(self.bias_off, dtype=np.int64)
            self.bias_log = np.ascontiguousarray(self.bias_log, dtype=np.float64)
        if self.seq is not None:
            if self.bias_log is not None:
                raise ValueError("give either bias_log or seq, not both")
            self.seq_off = np.ascontiguousarray(self.seq_off, dtype=np.int64)
            self.seq = np.ascontiguousarray(self.seq, dtype=np.uint8)
            self.pwm_log = np.ascontiguousarray(self.pwm_log, dtype=np.float64)
            self.pwm_nucs = np.ascontiguousarray(self.pwm_nucs, dtype=np.uint8)
        if self.out_off is None:
            self.out_off = np.zeros(self.n_chunks + 1, dtype=np.int64)
            np.cumsum(self.chunk_len, out=self.out_off[1:])
        self.validate()

    @property
    def n_chunks(self):
        return int(self.chunk_len.shape[0])

    @property
    def n_frags(self):
        return int(self.frag_lpos.shape[0])

    @property
    def total_bp(self):
        return int(self.out_off[-1])

    def validate(self):
        nc = self.n_chunks
        if self.frag_off.shape[0] != nc + 1 or self.frag_off[0] != 0 or self.frag_off[-1] != self.n_frags:
            raise ValueError("frag_off must be CSR offsets of length n_chunks+1")
        if np.any(np.diff(self.frag_off) < 0):
            raise ValueError("frag_off must be non-decreasing")
        if self.frag_ilen.shape != self.frag_lpos.shape:
            raise ValueError("frag_lpos / frag_ilen length mismatch")
        if self.bias_log is not None:
            want = self.chunk_len.astype(np.int64) + self.bias_left + self.bias_right
            if self.bias_off.shape[0] != nc + 1 or np.any(np.diff(self.bias_off) != want):
                raise ValueError("bias_log must cover [start-bias_left, end+bias_right) for every chunk")
        if self.seq is not None:
            K = self.pwm_log.shape[1]
            want = self.chunk_len.astype(np.int64) + self.bias_left + self.bias_right + K - 1
            if self.seq_off.shape[0] != nc + 1 or np.any(np.diff(self.seq_off) != want) or self.seq_off[-1] != self.seq.shape[0]:
                raise ValueError("seq must hold [start-bias_left-up, end+bias_right+down) for every chunk")
        if np.any(self.chunk_len <= 0):
            raise ValueError("empty chunk")

    def chunk_frags(self, i):
        a, b = int(self.frag_off[i]), int(self.frag_off[i + 1])
        return self.frag_lpos[a:b], self.frag_ilen[a:b]

    def chunk_bias(self, i):
        if self.bias_log is None:
            return None
        return self.bias_log[int(self.bias_off[i]):int(self.bias_off[i + 1])]

    def subset(self, lo, hi):
        """contiguous chunk range [lo, hi) as a new PackedChunks (used to shard over GPUs)."""
        fa, fb = int(self.frag_off[lo]), int(self.frag_off[hi])
        kw = dict(chunk_start=self.chunk_start[lo:hi], chunk_len=self.chunk_len[lo:hi],
                  frag_off=self.frag_off[lo:hi + 1] - fa, frag_lpos=self.frag_lpos[fa:fb],
                  frag_ilen=self.frag_ilen[fa:fb], bias_off=None, bias_log=None,
                  bias_left=self.bias_left, bias_right=self.bias_right,
                  chroms=self.chroms[lo:hi] if self.chroms else [])
        if self.bias_log is not None:
            ba, bb = int(self.bias_off[lo]), int(self.bias_off[hi])
            kw["bias_off"] = self.bias_off[lo:hi + 1] - ba
            kw["bias_log"] = self.bias_log[ba:bb]
        if self.seq is not None:
            sa, sb = int(self.seq_off[lo]), int(self.seq_off[hi])
            kw.update(seq_off=self.seq_off[lo:hi + 1] - sa, seq=self.seq[sa:sb], pwm_log=self.pwm_log, pwm_nucs=self.pwm_nucs)
        return PackedChunks(**kw)


def sort_by_centre(lpos, ilen):
    """stable order of fragments by centre l + (n-1)//2 (pyatac/fragments.pyx:36)."""
    c = lpos.astype(np.int64) + (ilen.astype(np.int64) - 1) // 2
    return np.argsort(c, kind="stable")


def pack_chunks(chunks, frag_l, frag_n, frag_chrom_off=None, bias_tracks=None, margin=None,
                bias_left=BIAS_LEFT, bias_right=BIAS_RIGHT):
    """Build PackedChunks from per-chromosome fragment arrays.

    chunks      : list of (chrom, start, end)
    frag_l/n    : dict chrom -> int64 arrays (absolute l, n), l sorted ascending
    bias_tracks : optional dict chrom -> (track_start, float64 log-bias array)
    margin      : fragments with l in [start - margin, end + margin) are attached to a chunk
                  (superset of the reference's fetch window, pyatac/fragments.pyx:24).
    """
    if margin is None:
        margin = 2000 + 126
    starts, lens, offs, ls, ns, boffs, bvals, chroms = [], [], [0], [], [], [0], [], []
    for chrom, s, e in chunks:
        L, N = frag_l[chrom], frag_n[chrom]
        a = int(np.searchsorted(L, s - margin, "left"))
        b = int(np.searchsorted(L, e + margin, "left"))
        l_rel = (L[a:b] - s).astype(np.int32)
        n_ = N[a:b].astype(np.int32)
        o = sort_by_centre(l_rel, n_)
        ls.append(l_rel[o])
        ns.append(n_[o])
        offs.append(offs[-1] + (b - a))
        starts.append(s)
        lens.append(e - s)
        chroms.append(chrom)
        if bias_tracks is not None:
            t0, vals = bias_tracks[chrom]
            x0, x1 = s - bias_left - t0, e + bias_right - t0
            if x0 < 0 or x1 > len(vals):
                raise ValueError("bias track does not cover chunk %s:%d-%d" % (chrom, s, e))
            bvals.append(vals[x0:x1])
            boffs.append(boffs[-1] + (x1 - x0))
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    return PackedChunks(chunk_start=np.array(starts, np.int64), chunk_len=np.array(lens, np.int32),
                        frag_off=np.array(offs, np.int64), frag_lpos=cat(ls, np.int32), frag_ilen=cat(ns, np.int32),
                        bias_off=np.array(boffs, np.int64) if bias_tracks is not None else None,
                        bias_log=cat(bvals, np.float64) if bias_tracks is not None else None,
                        bias_left=bias_left, bias_right=bias_right, chroms=chroms)
