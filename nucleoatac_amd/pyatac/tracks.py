"""Per-base signal tracks (API of the reference's pyatac/tracks.py:16-222)."""
import gzip
import os
import threading

import numpy as np

from .chunk import Chunk
from .utils import smooth


_LOCAL = threading.local()


def _tabix(path):
    """one open index per file and thread (NucChunk.getOcc reads three tracks per chunk, NucleosomeCalling.py:284-293; the
    batched driver reads the chunks of a batch from several threads and a reader holds one file position): the native reader
    when libnatac_hip.so is there, else the pure-Python one"""
    from ..tabix import NativeTabix, TabixFile
    cache = _LOCAL.__dict__.setdefault("tabix", {})
    key = (path, os.path.getmtime(path + ".tbi"))
    tb = cache.get(path)
    if tb is None or tb[0] != key:
        if tb is not None:
            tb[1].close()
        try:
            rd = NativeTabix(path)
        except ImportError:
            rd = TabixFile(path)
        tb = (key, rd)
        cache[path] = tb
    return tb[1]


def _py2_float_str(v):
    """python-2 str(float): 12 significant digits (what the reference's text outputs contain)"""
    if isinstance(v, (float, np.floating)):
        s = "%.12g" % v
        if "." not in s and "e" not in s and "n" not in s and "i" not in s:
            s += ".0"
        return s
    return str(v)


class Track(Chunk):
    """generic signal track over [start, end) (pyatac/tracks.py:16-156)"""

    def __init__(self, chrom, start, end, name="track", vals=None, log=False):
        Chunk.__init__(self, chrom, start, end, name=name)
        self.log = log
        if vals is None:
            self.vals = None
        elif len(vals) == self.length():
            self.vals = vals
        else:
            raise Exception("Input vals must be of length as set by start and end!")

    def assign_track(self, vals, start=None, end=None):
        if start:
            self.start = start
        if end:
            self.end = end
        if len(vals) != self.end - self.start:
            raise Exception("The values being assigned to track do not span the start to end of the track")
        self.vals = vals

    def write_track(self, handle, start=None, end=None, vals=None, write_zero=True, keep_runs_before_nan=False):
        """run-length bedGraph text, NaN runs skipped (pyatac/tracks.py:37-74).  Like the reference, whose loop overwrites
        `prev_value` with a NaN before flushing the open run (tracks.py:56-66), a run of values that is directly followed
        by a NaN is NOT written; keep_runs_before_nan=True writes those runs too (deviation)."""
        if start is None:
            start = self.start
        if end is None:
            end = self.end
        if vals is None:
            vals = self.vals
        if len(vals) != self.end - self.start:
            raise Exception("Error! Inconsistency between length of values and start/end values")
        vals = np.asarray(vals, dtype=np.float64)
        n = len(vals)
        if n == 0:
            return
        nan = np.isnan(vals)
        # run boundaries: value changes (NaN != NaN handled through the mask)
        change = np.ones(n, dtype=bool)
        change[1:] = (vals[1:] != vals[:-1]) & ~(nan[1:] & nan[:-1])
        starts = np.flatnonzero(change)
        ends = np.append(starts[1:], n)
        out = []
        for a, b in zip(starts, ends):
            v = vals[a]
            if nan[a] or (v == 0 and not write_zero):
                continue
            if b < n and nan[b] and not keep_runs_before_nan:
                continue
            out.append("%s\t%d\t%d\t%s\n" % (self.chrom, start + a, start + b, _py2_float_str(float(v))))
        handle.write("".join(out))

    def read_track(self, bedgraph, start=None, end=None, empty=np.nan, flank=None):
        """read values from a (gzipped) bedGraph file (pyatac/tracks.py:75-87): through the tabix index when
        `bedgraph`.tbi exists (pysam.TabixFile.fetch of the reference), else by a linear scan"""
        if start:
            self.start = start
        if end:
            self.end = end
        if flank:
            self.start -= flank
            self.end += flank
        out = np.ones(self.end - self.start) * empty
        if bedgraph.endswith(".gz") and os.path.exists(bedgraph + ".tbi"):
            rd = _tabix(bedgraph)
            if hasattr(rd, "read_values"):
                self.vals = rd.read_values(self.chrom, self.start, self.end, empty=empty)
                return
            b0, e0, v0 = rd.fetch_values(self.chrom, max(0, self.start), self.end)
            n = self.end - self.start
            a = np.clip(b0 - self.start, 0, n)
            z = np.clip(e0 - self.start, 0, n)
            single = (z - a) == 1
            out[a[single]] = v0[single]                      # one line per base is the common case of a float track
            for i in np.nonzero(~single)[0]:                 # in file order, like the reference's line loop
                out[a[i]:z[i]] = v0[i]
            self.vals = out
            return
        opener = gzip.open if bedgraph.endswith(".gz") else open
        with opener(bedgraph, "rt") as fh:
            for line in fh:
                f = line.rstrip("\n").split("\t")
                if f[0] != self.chrom:
                    continue
                s, e = int(f[1]), int(f[2])
                if e > self.start and s < self.end:
                    out[max(s - self.start, 0):min(e - self.start, self.end - self.start)] = float(f[3])
        self.vals = out

    def exp(self):
        self.vals = np.exp(self.vals)
        self.log = False

    def smooth_track(self, window_len, window="flat", sd=None, mode="valid", norm=True):
        """smooth in place; 'valid' shrinks the interval by window_len//2 per side (pyatac/tracks.py:101-109)"""
        self.smoothed = True
        self.vals = smooth(self.vals, window_len, window=window, sd=sd, mode=mode, norm=norm)
        if mode == "valid":
            self.start = self.start + window_len // 2
            self.end = self.end - window_len // 2

    def get(self, start=None, end=None, pos=None):
        if pos:
            try:
                return self.vals[pos - self.start]
            except Exception:
                raise Exception("Looks like position given doesn't match track")
        if start is None:
            start = self.start
        if end is None:
            end = self.end
        try:
            return self.vals[start - self.start:end - self.start]
        except Exception:
            raise Exception("Looks like dimensions from get probaby don't match track, or there are no vals in track")

    def slop(self, chromDict, up=0, down=0, new=False):
        if self.vals is None:
            return Chunk.slop(self, chromDict, up=up, down=down, new=new)
        raise Exception("Cannot slop Track if vals are set")


class InsertionTrack(Track):
    """Tn5 insertion counts (pyatac/tracks.py:160-201)"""

    def __init__(self, chrom, start, end):
        Track.__init__(self, chrom, start, end, "insertions")

    def calculateInsertions(self, bamfile, flank=0, lower=0, upper=2000, atac=True):
        from .fragments import getInsertions
        self.start -= flank
        self.end += flank
        self.vals = getInsertions(bamfile, self.chrom, self.start, self.end, lower, upper, atac)

    def calculateStrandedInsertions(self, bamfile, flank=0, lower=0, upper=2000, atac=True):
        from .fragments import getStrandedInsertions
        self.start -= flank
        self.end += flank
        self.plus, self.minus = getStrandedInsertions(bamfile, self.chrom, self.start, self.end, lower, upper, atac)
        self.vals = self.plus + self.minus


class CoverageTrack(Track):
    """fragment-centre coverage in a flat window (pyatac/tracks.py:204-222)"""

    def __init__(self, chrom, start, end):
        Track.__init__(self, chrom, start, end, "coverage")

    def calculateCoverage(self, mat, lower, upper, window_len):
        offset = self.start - mat.start - (window_len // 2)
        if offset < 0:
            raise Exception("Insufficient flanking region on mat to calculate coverage with desired window")
        lo, up = lower - mat.lower, upper - mat.lower
        sub = mat.mat[lo:up, offset:mat.mat.shape[1] - offset] if offset else mat.mat[lo:up]
        self.vals = smooth(np.sum(sub, axis=0), window_len, window="flat", mode="valid", norm=False)

    def calculateCoverageSmooth(self, mat, lower, upper, window_len, sd):
        offset = self.start - mat.start - (window_len // 2)
        if offset < 0:
            raise Exception("Insufficient flanking region on mat to calculate coverage with desired window")
        lo, up = lower - mat.lower, upper - mat.lower
        sub = mat.mat[lo:up, offset:mat.mat.shape[1] - offset] if offset else mat.mat[lo:up]
        self.vals = smooth(np.sum(sub, axis=0), window_len, sd=sd, window="gaussian", mode="valid", norm=False)
