"""Reader for tabix-indexed BGZF BED / bedGraph files (the role of pysam.TabixFile.fetch in pyatac/tracks.py:75-87
and pyatac/bias.py).  Pure host code: the index is written by natac_tabix_index (csrc/natac_tabix.hpp)."""
import gzip
import struct
import zlib

import numpy as np


def reg2bins(beg, end):
    """bins of the 5-level / 16 kb binning scheme that may hold records overlapping [beg, end)"""
    end -= 1
    bins = [0]
    for shift, off in ((26, 1), (23, 9), (20, 73), (17, 585), (14, 4681)):
        bins.extend(range(off + (beg >> shift), off + (end >> shift) + 1))
    return bins


class TabixFile(object):
    def __init__(self, path, index=None):
        self.path = path
        with gzip.open(index or path + ".tbi", "rb") as fh:
            raw = fh.read()
        if raw[:4] != b"TBI\x01":
            raise ValueError("not a tabix index: %s" % (index or path + ".tbi"))
        n_ref, self.format, self.col_seq, self.col_beg, self.col_end, self.meta, self.skip, l_nm = struct.unpack_from("<8i", raw, 4)
        p = 36
        self.contigs = [n.decode() for n in raw[p:p + l_nm].split(b"\0")[:-1]]
        p += l_nm
        self.bins, self.lin, self.stats = [], [], []
        for _ in range(n_ref):
            (n_bin,) = struct.unpack_from("<i", raw, p)
            p += 4
            bins = {}
            for _b in range(n_bin):
                b, n_chunk = struct.unpack_from("<Ii", raw, p)
                p += 8
                bins[b] = [struct.unpack_from("<QQ", raw, p + 16 * i) for i in range(n_chunk)]
                p += 16 * n_chunk
            self.stats.append(bins.pop(37450, None))
            (n_intv,) = struct.unpack_from("<i", raw, p)
            p += 4
            self.lin.append(struct.unpack_from("<%dQ" % n_intv, raw, p))
            p += 8 * n_intv
            self.bins.append(bins)
        self._fh = open(path, "rb")

    def close(self):
        self._fh.close()

    def _block(self, coff):
        self._fh.seek(coff)
        head = self._fh.read(18)
        if len(head) < 18:
            return b"", 0
        xlen = struct.unpack_from("<H", head, 10)[0]
        extra = head[12:] + self._fh.read(xlen - 6)
        bsize, q = None, 0
        while q + 4 <= len(extra):
            slen = struct.unpack_from("<H", extra, q + 2)[0]
            if extra[q:q + 2] == b"BC":
                bsize = struct.unpack_from("<H", extra, q + 4)[0] + 1
            q += 4 + slen
        body = self._fh.read(bsize - 12 - xlen)
        return zlib.decompress(body[:-8], -15), bsize

    def _read(self, v0, v1):
        """inflated bytes between two virtual offsets"""
        out = []
        coff, uoff = v0 >> 16, v0 & 0xffff
        while coff < (v1 >> 16) or (coff == (v1 >> 16) and uoff < (v1 & 0xffff)):
            data, bsize = self._block(coff)
            if bsize == 0:
                break
            stop = (v1 & 0xffff) if coff == (v1 >> 16) else len(data)
            out.append(data[uoff:stop])
            coff, uoff = coff + bsize, 0
        return b"".join(out)

    def fetch(self, chrom, start, end):
        """lines (str, without newline) of records on `chrom` overlapping [start, end)"""
        if chrom not in self.contigs:
            return
        tid = self.contigs.index(chrom)
        start = max(0, int(start))
        end = int(end)
        if end <= start:
            return
        lin = self.lin[tid]
        w = start >> 14
        min_off = lin[w] if w < len(lin) else (lin[-1] if lin else 0)
        chunks = []
        for b in reg2bins(start, end):
            for c0, c1 in self.bins[tid].get(b, ()):
                if c1 > min_off:
                    chunks.append((max(c0, min_off), c1))
        chunks.sort()
        merged = []
        for c0, c1 in chunks:
            if merged and c0 <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], c1)
            else:
                merged.append([c0, c1])
        for c0, c1 in merged:
            for line in self._read(c0, c1).split(b"\n"):
                if not line or line[:1] == b"#":
                    continue
                f = line.split(b"\t")
                if f[self.col_seq - 1].decode() != chrom:
                    continue
                b0 = int(f[self.col_beg - 1])
                e0 = max(int(f[self.col_end - 1]), b0 + 1)
                if b0 >= end:
                    return
                if e0 > start:
                    yield line.decode()

    def _before(self, other, name):
        """chromosome `other` (bytes) precedes `name` in the file's order"""
        o = other.decode()
        return o in self.contigs and self.contigs.index(o) < self.contigs.index(name.decode())

    def fetch_values(self, chrom, start, end, value_col=4):
        """(beg, end, value) arrays of the records on `chrom` overlapping [start, end) -- the bedGraph case of fetch(),
        parsed in bulk (Track.read_track reads ~one line per base)"""
        empty = (np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float64))
        if chrom not in self.contigs:
            return empty
        tid = self.contigs.index(chrom)
        start = max(0, int(start))
        end = int(end)
        if end <= start:
            return empty
        lin = self.lin[tid]
        w = start >> 14
        min_off = lin[w] if w < len(lin) else (lin[-1] if lin else 0)
        chunks = []
        for b in reg2bins(start, end):
            for c0, c1 in self.bins[tid].get(b, ()):
                if c1 > min_off:
                    chunks.append((max(c0, min_off), c1))
        chunks.sort()
        merged = []
        for c0, c1 in chunks:
            if merged and c0 <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], c1)
            else:
                merged.append([c0, c1])
        bs, es, vs = [], [], []
        name = chrom.encode()
        cb, ce, cs = self.col_beg - 1, self.col_end - 1, self.col_seq - 1
        ncol = max(cb, ce, cs, value_col - 1) + 1
        for c0, c1 in merged:
            lines = [ln for ln in self._read(c0, c1).split(b"\n") if ln and ln[:1] != b"#"]
            # a chunk holds whole bins (>= 16 kb of records, position-sorted): bisect on the begin column instead of parsing
            # every line.  bedGraph records do not overlap, so the first overlapping record is at most a few lines before
            # the first one that begins at or after `start`; the backward scan is bounded and checked.
            lo, hi = 0, len(lines)
            while lo < hi:                                   # first line of this chromosome that begins at or after `end`
                mid = (lo + hi) >> 1
                f = lines[mid].split(b"\t", ncol)
                if f[cs] == name and int(f[cb]) < end or f[cs] != name and mid < len(lines) and self._before(f[cs], name):
                    lo = mid + 1
                else:
                    hi = mid
            stop = lo
            lo, hi = 0, stop
            while lo < hi:                                   # first line of this chromosome that begins at or after `start`
                mid = (lo + hi) >> 1
                f = lines[mid].split(b"\t", ncol)
                if f[cs] == name and int(f[cb]) < start or f[cs] != name and self._before(f[cs], name):
                    lo = mid + 1
                else:
                    hi = mid
            first = lo
            back = 0
            while first > 0 and back < 64:
                f = lines[first - 1].split(b"\t", ncol)
                if f[cs] != name:
                    break
                if max(int(f[ce]), int(f[cb]) + 1) > start:
                    first -= 1
                    back = 0
                else:
                    back += 1
                    if back >= 8:
                        break
                    first -= 1
            keep = [c for c in (ln.split(b"\t", ncol) for ln in lines[first:stop]) if c[cs] == name]
            if not keep:
                continue
            b0 = np.array([c[cb] for c in keep]).astype(np.int64)
            e0 = np.array([c[ce] for c in keep]).astype(np.int64)
            v0 = np.array([c[value_col - 1] for c in keep]).astype(np.float64)
            m = (b0 < end) & (np.maximum(e0, b0 + 1) > start)
            bs.append(b0[m]); es.append(e0[m]); vs.append(v0[m])
        if not bs:
            return empty
        return np.concatenate(bs), np.concatenate(es), np.concatenate(vs)


class NativeTabix(object):
    """the same region reads through libnatac_hip.so (natac_tbx_*, csrc/natac_tabix.hpp): ~20x faster than the parser above"""

    def __init__(self, path):
        import ctypes as C
        from . import _lib as L
        self._L, self._C = L, C
        self._lib = L.load()
        self._h = C.c_void_p()
        L.check(self._lib.natac_tbx_open(str(path).encode(), C.byref(self._h)))

    def read_values(self, chrom, start, end, empty=np.nan, value_col=4):
        C = self._C
        out = np.empty(max(0, int(end) - int(start)), np.float64)
        n = C.c_int64(0)
        self._L.check(self._lib.natac_tbx_read_values(self._h, str(chrom).encode(), int(start), int(end), int(value_col),
                                                      float(empty), out.ctypes.data_as(C.c_void_p), C.byref(n)))
        return out

    def read_regions(self, chroms, starts, ends, empty=np.nan, value_col=4, n_threads=0):
        """read_values of many regions in one call (natac_tbx_read_regions): (flat values, offsets); region i is
        flat[off[i]:off[i + 1]].  Contiguous runs of the list go to the handle's cursors, so a position-sorted list inflates
        and parses every BGZF member once."""
        C = self._C
        names = sorted(set(chroms))
        idx = {c: i for i, c in enumerate(names)}
        cid = np.array([idx[c] for c in chroms], dtype=np.int32)
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        ends = np.ascontiguousarray(ends, dtype=np.int64)
        off = np.concatenate(([0], np.cumsum(np.maximum(ends - starts, 0)))).astype(np.int64)
        out = np.empty(int(off[-1]), np.float64)
        arr = (C.c_char_p * max(1, len(names)))(*[str(c).encode() for c in names])
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        n = C.c_int64(0)
        self._L.check(self._lib.natac_tbx_read_regions(self._h, len(cid), vp(cid), arr, len(names), vp(starts), vp(ends), int(value_col),
                                                       float(empty), vp(out), vp(off), int(n_threads), C.byref(n)))
        return out, off

    def close(self):
        if self._h:
            self._lib.natac_tbx_close(self._h)
            self._h = None

    def __del__(self):          # readers cached per thread go away with their thread
        try:
            self.close()
        except Exception:
            pass
