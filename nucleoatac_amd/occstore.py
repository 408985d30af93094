"""Occupancy tracks kept in HBM between the stages of one process (`nucleoatac run`).

The reference chains occ -> nuc -> nfr through files (cli.py:34-64): `occ` writes three .bedgraph.gz tracks, `nuc` reads them back
per chunk (NucChunk.getOcc, NucleosomeCalling.py:284-293) and so does `nfr` (NFRChunk.getOcc, NFRCalling.py:64-67).  The files are
still written here; but when the reader runs in the process that wrote them, the values come out of a device-resident copy that
holds exactly what the file shows -- every run rounded to its twelve printed digits, NaN where Track.write_track writes no line
(natac_store_adopt) -- so the outputs are byte-identical to the file path and no occupancy text is inflated or parsed.

A store is registered under the path of the `occ` track file; readers ask `lookup(path)`.  A region that the store does not cover
(another rank's shard, a sub-batch the device could not round exactly) makes `read_regions` return None and the caller reads the
file, as before."""
import os
import threading

import numpy as np

from . import _lib as L
from .device import TrackStore

_REGISTRY = {}
_LOCK = threading.Lock()
SLOTS = {"occ": 0, "lower_bound": 1, "upper_bound": 2}        # slot of a track inside a segment = order of TRACKS
TRACKS = (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER)
ENABLED = os.environ.get("NATAC_RESIDENT_OCC", "1") != "0"     # NATAC_RESIDENT_OCC=0: always through the files (A/B, validation)


class OccTrackStore(object):
    """genome-interval index over the segments of a device TrackStore: chunk (chrom, start, end) -> (segment, offset)"""

    def __init__(self):
        self.dev = TrackStore()
        self._rows = {}            # chrom -> list of (start, end, segment, offset); sorted lazily
        self._sorted = {}
        self.reads = 0             # regions served (tests / phase reports)

    def add(self, chunks, out_off, segment):
        """the chunks of one adopted sub-batch: chunk k = values [out_off[k], out_off[k + 1]) of `segment`"""
        if segment is None or segment < 0:
            return
        for k, c in enumerate(chunks):
            self._rows.setdefault(c.chrom, []).append((int(c.start), int(c.end), int(segment), int(out_off[k])))
        self._sorted = {}

    def _table(self, chrom):
        t = self._sorted.get(chrom)
        if t is None:
            rows = sorted(self._rows.get(chrom, []))
            t = self._sorted[chrom] = tuple(np.array([r[i] for r in rows], dtype=np.int64) for i in range(4))
        return t

    def locate(self, chroms, starts, ends):
        """(segment, offset) arrays for regions [start, end), or None unless EVERY region lies inside one stored chunk"""
        n = len(chroms)
        seg, off = np.empty(n, np.int64), np.empty(n, np.int64)
        starts, ends = np.asarray(starts, dtype=np.int64), np.asarray(ends, dtype=np.int64)
        by = {}
        for i, c in enumerate(chroms):
            by.setdefault(c, []).append(i)
        for c, idx in by.items():
            s0, e0, sg, of = self._table(c)
            if not len(s0):
                return None
            idx = np.array(idx)
            j = np.searchsorted(s0, starts[idx], "right") - 1
            if (j < 0).any() or (ends[idx] > e0[np.maximum(j, 0)]).any():
                return None
            seg[idx], off[idx] = sg[j], of[j] + (starts[idx] - s0[j])
        return seg, off

    def read_regions(self, ctx, chroms, starts, ends, slot):
        """(flat values, offsets) of the regions from slot `slot` -- the shape pyatac.tracks' native reader returns --, or None"""
        loc = self.locate(chroms, starts, ends)
        if loc is None:
            return None
        lens = np.asarray(ends, dtype=np.int64) - np.asarray(starts, dtype=np.int64)
        from . import context_lock
        with context_lock:
            flat = self.dev.read(ctx, loc[0], loc[1], lens, slot)
        self.reads += len(lens)
        return flat, np.concatenate(([0], np.cumsum(lens))).astype(np.int64)

    def read_points(self, ctx, chroms, positions, slot):
        """values at single positions (nuc's calls)"""
        loc = self.locate(chroms, positions, np.asarray(positions, dtype=np.int64) + 1)
        if loc is None:
            return None
        self.reads += len(positions)
        from . import context_lock
        with context_lock:
            return self.dev.read(ctx, loc[0], loc[1], np.ones(len(positions), dtype=np.int64), slot)

    def close(self):
        self.dev.close()
        self._rows, self._sorted = {}, {}


def register(occ_track_path, store):
    with _LOCK:
        old = _REGISTRY.pop(os.path.abspath(occ_track_path), None)
        _REGISTRY[os.path.abspath(occ_track_path)] = store
    if old is not None and old is not store:
        old.close()


def lookup(occ_track_path):
    """the store whose `occ` track file this is, or None"""
    if not ENABLED or occ_track_path is None:
        return None
    with _LOCK:
        return _REGISTRY.get(os.path.abspath(occ_track_path))


def slot_of(path):
    """which of the three tracks a file name is: <out>.occ.bedgraph.gz / .occ.lower_bound.bedgraph.gz / .occ.upper_bound.bedgraph.gz"""
    base = os.path.basename(path)
    if base.endswith(".occ.lower_bound.bedgraph.gz"):
        return SLOTS["lower_bound"], path[:-len("lower_bound.bedgraph.gz")] + "bedgraph.gz"
    if base.endswith(".occ.upper_bound.bedgraph.gz"):
        return SLOTS["upper_bound"], path[:-len("upper_bound.bedgraph.gz")] + "bedgraph.gz"
    return SLOTS["occ"], path


def release(occ_track_path=None):
    """drop one store (or all): frees the HBM it holds"""
    with _LOCK:
        keys = [os.path.abspath(occ_track_path)] if occ_track_path is not None else list(_REGISTRY)
        stores = [_REGISTRY.pop(k) for k in keys if k in _REGISTRY]
    for s in stores:
        s.close()
