"""Nucleosome-free regions between neighbouring nucleosome calls (API of the reference's nucleoatac/NFRCalling.py).

The per-base inputs come from the accelerated path: the insertion track of every chunk is natac_run_ins on one batch
(run_nfr.py hands the slice in), the bias track is the PWM kernel behind InsertionBiasTrack.computeBias, the occupancy tracks
are read back through the native tabix reader.  What is left here is interval bookkeeping: the gaps between consecutive
dyads, four statistics per gap, two thresholds."""
import numpy as np

from ..pyatac.bias import PWM, InsertionBiasTrack
from ..pyatac.chunk import Chunk
from ..pyatac.tracks import InsertionTrack, Track, _py2_float_str
from ..pyatac.utils import read_chrom_sizes_from_fasta
from ..tabix import TabixFile

DYAD_LEFT, DYAD_RIGHT = 73, 72      # an NFR starts 73 bp right of a dyad and ends 72 bp left of the next (NFRCalling.py:103-104)


def gap_statistics(nfrtrack, left, right):
    """(mean occupancy, min upper bound, mean insertions, mean bias) over [left, right) -- NFRCalling.py:23-26.  The
    reference needs --fasta for the last one (its bias track has no values otherwise and NFR() raises); here it is nan."""
    occ = float(np.mean(nfrtrack.occ.get(left, right)))
    min_upper = float(np.min(nfrtrack.occ_upper.get(left, right)))
    ins = float(np.mean(nfrtrack.ins.get(left, right)))
    bias = float(np.mean(nfrtrack.bias.get(left, right, log=False))) if nfrtrack.bias.vals is not None else float("nan")
    return occ, min_upper, ins, bias


class NFR(Chunk):
    """one nucleosome-free region with its four statistics (NFRCalling.py:16-32)"""

    def __init__(self, left, right, nfrtrack):
        self.chrom, self.start, self.end, self.strand = nfrtrack.chrom, left, right, "*"
        self.occ, self.min_upper, self.ins_density, self.bias_density = gap_statistics(nfrtrack, left, right)

    def passes(self, params):
        """depleted of nucleosomes: NaN statistics fail both comparisons, as in the reference (NFRCalling.py:108)"""
        return self.min_upper < params.max_occ_upper and self.occ < params.max_occ

    def asBed(self):
        cols = [self.chrom, self.start, self.end]
        cols += [_py2_float_str(v) for v in (self.occ, self.min_upper, self.ins_density, self.bias_density)]
        return "\t".join(str(c) for c in cols)

    def write(self, handle):
        handle.write(self.asBed() + "\n")


class NFRParameters(object):
    """run-level settings of `nucleoatac nfr` (NFRCalling.py:35-48)"""

    def __init__(self, occ_track, calls, ins_track=None, bam=None, max_occ=0.25, max_occ_upper=0.25, fasta=None, pwm=None):
        self.occ_track, self.calls, self.ins_track, self.bam = occ_track, calls, ins_track, bam
        self.max_occ, self.max_occ_upper = max_occ, max_occ_upper
        self.fasta = fasta
        if fasta is not None:
            self.pwm = PWM.open(pwm)
            self.chrs = read_chrom_sizes_from_fasta(fasta)

    def upper_bound_track(self):
        """<out>.occ.bedgraph.gz -> <out>.occ.upper_bound.bedgraph.gz (the reference strips the last 11 characters, :66)"""
        return self.occ_track[:-len("bedgraph.gz")] + "upper_bound.bedgraph.gz"


class NFRChunk(Chunk):
    """NFR calls of one region (NFRCalling.py:52-118)"""

    def __init__(self, chunk):
        self.chrom, self.start, self.end = chunk.chrom, chunk.start, chunk.end
        self.nfrs = []

    def initialize(self, parameters):
        self.params = parameters

    def _read(self, path, name):
        t = Track(self.chrom, self.start, self.end, name)
        t.read_track(path)
        return t

    def getOcc(self):
        self.occ = self._read(self.params.occ_track, "Occupancy")
        self.occ_upper = self._read(self.params.upper_bound_track(), "Occupancy")

    def getIns(self, vals=None):
        """insertion track: the slice of the GPU batch (`vals`), or this region alone through getInsertions, or --ins_track"""
        if self.params.ins_track is not None:
            self.ins = self._read(self.params.ins_track, "Insertion")
            return
        self.ins = InsertionTrack(self.chrom, self.start, self.end)
        if vals is None:
            self.ins.calculateInsertions(self.params.bam)
        else:
            self.ins.vals = vals

    def getBias(self):
        self.bias = InsertionBiasTrack(self.chrom, self.start, self.end, log=True)
        if self.params.fasta is not None:
            self.bias.computeBias(self.params.fasta, self.params.chrs, self.params.pwm)

    def dyads(self):
        """positions of the combined calls overlapping the region, in file order"""
        tbx = TabixFile(self.params.calls)
        try:
            if self.chrom not in tbx.contigs:
                return []
            return [int(row.split("\t")[1]) for row in tbx.fetch(self.chrom, self.start, self.end)]
        finally:
            tbx.close()

    def findNFRs(self):
        """gaps between consecutive dyads that are wide enough and depleted of nucleosomes (NFRCalling.py:96-110)"""
        nucs = self.dyads()
        gaps = [(a + DYAD_LEFT, b - DYAD_RIGHT) for a, b in zip(nucs[:-1], nucs[1:])]
        for left, right in gaps:
            if right > left:
                candidate = NFR(left, right, self)
                if candidate.passes(self.params):
                    self.nfrs.append(candidate)

    def process(self, params, ins_vals=None):
        self.initialize(params)
        self.getOcc()
        self.getIns(ins_vals)
        self.getBias()
        self.findNFRs()

    def removeData(self):
        self.__dict__.clear()


def nfr_batch(chunks, params, ins_off=None, ins_flat=None):
    """NFRChunk.process for a list of chunks with the per-chunk work batched: ONE PWM launch for the bias of all chunks
    (pipeline._bias_batch), the occupancy tracks and the dyad positions of the whole batch read with one native call per file
    (natac_tbx_read_regions: every BGZF member inflated and parsed once), then the interval statistics of NFR.__init__ with
    the reference's own numpy expressions per gap.  Returns (chunk index, left, right, values[n, 4]) of the NFRs that pass, in
    chunk / position order -- what the per-chunk loop writes."""
    from ..pipeline import _bias_batch
    from ..pyatac.tracks import _tabix
    from .NucleosomeCalling import map_in_slices, read_regions_of
    n = len(chunks)
    starts = np.array([c.start for c in chunks], dtype=np.int64)
    ends = np.array([c.end for c in chunks], dtype=np.int64)
    boff = bias = None
    if params.fasta is not None:
        boff, bias = _bias_batch(chunks, starts, ends, params.fasta, params.pwm, 0, 0)
    upper = params.upper_bound_track()

    def track(path, ch):             # Track.read_track: the native tabix reader when the file is indexed, else a linear scan
        t = Track(ch.chrom, ch.start, ch.end)
        t.read_track(path)
        return t.vals

    def column(path, value_col=4):   # (flat values, offsets) of one file: one native call for the batch, else chunk by chunk
        if value_col == 4:           # an occupancy track this process wrote: its values are still in HBM, as the file shows them
            from .. import get_context, occstore
            slot, occ_path = occstore.slot_of(path)
            res = occstore.lookup(occ_path)
            if res is not None:
                got = res.read_regions(get_context(), [c.chrom for c in chunks], starts, ends, slot)
                if got is not None:
                    return got
        got = read_regions_of(path, chunks, value_col)
        if got is not None:
            return got
        if value_col == 2:           # record starts (the dyads) through the line reader
            def starts_of(ch):
                rd = _tabix(path)
                out = np.full(ch.end - ch.start, np.nan)
                if ch.chrom in rd.contigs:
                    for row in rd.fetch(ch.chrom, ch.start, ch.end):
                        p = int(row.split("\t")[1])
                        if ch.start <= p < ch.end:
                            out[p - ch.start] = p
                return out
            parts = map_in_slices(starts_of, chunks)
        else:
            parts = map_in_slices(lambda ch: track(path, ch), chunks)
        return (np.concatenate(parts) if parts else np.zeros(0)), np.concatenate(([0], np.cumsum([len(p) for p in parts]))).astype(np.int64)

    (occ, off), (up, _), (dy, _) = column(params.occ_track), column(upper), column(params.calls, 2)   # calls: tabix-indexed like in the reference
    if params.ins_track is not None:
        ins, ioff = column(params.ins_track)
    elif ins_flat is not None:
        ins, ioff = ins_flat, np.asarray(ins_off, dtype=np.int64)
    else:
        parts = []
        for ch in chunks:
            t = InsertionTrack(ch.chrom, ch.start, ch.end)
            t.calculateInsertions(params.bam)
            parts.append(t.vals)
        ins, ioff = np.concatenate(parts) if parts else np.zeros(0), off
    # gaps between consecutive dyads of a chunk (NFRCalling.py:96-104), all chunks at once: the dyads are the non-NaN entries of `dy`
    at = np.flatnonzero(~np.isnan(dy))                       # flat indices of the dyads, ascending = chunk / position order
    ck = np.searchsorted(off, at, "right") - 1               # their chunks
    same = ck[1:] == ck[:-1]
    gk = ck[:-1][same]
    rel_l = (at[:-1] - off[ck[:-1]])[same] + DYAD_LEFT       # chunk-relative [left, right)
    rel_r = (at[1:] - off[ck[1:]])[same] - DYAD_RIGHT
    wide = rel_r > rel_l
    gk, rel_l, rel_r = gk[wide], rel_l[wide], rel_r[wide]
    glen = rel_r - rel_l

    def per_gap(flat, base, sel, fn):
        """fn over flat[base[g] + rel_l[g] : base[g] + rel_r[g]] for the gaps in `sel`: gaps of equal length are gathered into one
        2-D array and reduced along its rows -- the same pairwise sums / exact minima as numpy's 1-D reductions on each slice"""
        out = np.empty(len(sel))
        if len(sel) == 0:
            return out
        order = sel[np.argsort(glen[sel], kind="stable")]
        cuts = np.concatenate(([0], np.flatnonzero(np.diff(glen[order])) + 1, [len(order)]))
        for a, b in zip(cuts[:-1], cuts[1:]):
            g = order[a:b]
            rows = flat[(base[gk[g]] + rel_l[g])[:, None] + np.arange(glen[g[0]])]
            out[np.searchsorted(sel, g)] = fn(rows)
        return out

    everything = np.arange(len(gk))
    o = per_gap(occ, off, everything, lambda r: r.mean(axis=1))
    mu = per_gap(up, off, everything, lambda r: r.min(axis=1))
    with np.errstate(invalid="ignore"):
        ok = np.flatnonzero((mu < params.max_occ_upper) & (o < params.max_occ))   # NaN statistics fail both comparisons
    vals = np.empty((len(ok), 4))
    vals[:, 0], vals[:, 1] = o[ok], mu[ok]
    vals[:, 2] = per_gap(ins, ioff, ok, lambda r: r.mean(axis=1))
    vals[:, 3] = per_gap(bias, boff, ok, lambda r: np.exp(r).mean(axis=1)) if bias is not None else np.nan
    kc, lefts, rights = gk[ok], rel_l[ok] + starts[gk[ok]], rel_r[ok] + starts[gk[ok]]
    return kc.astype(np.int64), lefts.astype(np.int64), rights.astype(np.int64), vals
