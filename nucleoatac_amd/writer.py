"""Native bedGraph writer (natac_write_bedgraph): all tracks of a batch in one multi-threaded call."""
import ctypes as C

import numpy as np

from . import _lib as L


def write_bedgraph(path, chroms, chunk_start, out_off, vals, append=False, compress=0, finish=True, write_zero=True,
                   n_threads=0, keep_runs_before_nan=False):
    """run-length bedGraph text (Track.write_track semantics, pyatac/tracks.py:37-74, including its rule that a run of
    values directly followed by a NaN is not written) for many chunks at once.
    compress: 0 plain text, 1..9 BGZF at that deflate level.  Returns the number of bytes written."""
    lib = L.load()
    nc = len(chroms)
    chunk_start = np.ascontiguousarray(chunk_start, dtype=np.int64)
    out_off = np.ascontiguousarray(out_off, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    if len(chunk_start) != nc or len(out_off) != nc + 1 or (nc and out_off[-1] > len(vals)):
        raise ValueError("inconsistent chunk arrays")
    names = (C.c_char_p * nc)(*[str(c).encode("ascii") for c in chroms])
    nb = C.c_int64(0)
    L.check(lib.natac_write_bedgraph(str(path).encode(), 1 if append else 0, int(compress), 1 if finish else 0, nc, names,
                                     chunk_start.ctypes.data_as(C.c_void_p), out_off.ctypes.data_as(C.c_void_p),
                                     vals.ctypes.data_as(C.c_void_p), (1 if write_zero else 0) | (2 if keep_runs_before_nan else 0), int(n_threads), C.byref(nb)))
    return nb.value


def write_bed_rows(path, names, chrom_id, start, end, vals, append=True, labels=None, label_id=None):
    """rows `chrom start end v0 v1 ...` with python-2 float text (natac_write_bed_rows): OccPeak.asBed / Nucleosome.asBed lines
    for whole batches at once.  names: list of chromosome names, chrom_id: index into it per row, vals: (n_rows, n_cols);
    labels / label_id: one more text column at the end (MergedNuc.asBed's source)."""
    lib = L.load()
    chrom_id = np.ascontiguousarray(chrom_id, dtype=np.int32)
    start = np.ascontiguousarray(start, dtype=np.int64)
    end = np.ascontiguousarray(end, dtype=np.int64)
    vals = np.ascontiguousarray(vals, dtype=np.float64).reshape(len(chrom_id), -1)
    if not (len(start) == len(end) == len(chrom_id)):
        raise ValueError("inconsistent row arrays")
    arr = (C.c_char_p * max(1, len(names)))(*[str(c).encode("ascii") for c in names])
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    if labels is not None:
        label_id = np.ascontiguousarray(label_id, dtype=np.int32)
        larr = (C.c_char_p * max(1, len(labels)))(*[str(c).encode("ascii") for c in labels])
        L.check(lib.natac_write_bed_rows_labeled(str(path).encode(), 1 if append else 0, len(chrom_id), vp(chrom_id), arr, len(names),
                                                 vp(start), vp(end), vp(vals), vals.shape[1], vp(label_id), larr, len(labels)))
        return
    L.check(lib.natac_write_bed_rows(str(path).encode(), 1 if append else 0, len(chrom_id), vp(chrom_id), arr, len(names),
                                     vp(start), vp(end), vp(vals), vals.shape[1]))


def read_bed_table(path, cols):
    """(names, chrom_id, start, end, vals[n_rows, len(cols)]) of a (gzipped) BED-like file, parsed natively (natac_bedtab_*): column 0
    = chromosome, 1 / 2 = start / end, `cols` = 0-based columns read as float64 with python's float() rounding"""
    lib = L.load()
    c = np.ascontiguousarray(cols, dtype=np.int32)
    h = C.c_void_p()
    L.check(lib.natac_bedtab_open(str(path).encode(), c.ctypes.data_as(C.c_void_p), len(c), C.byref(h)))
    try:
        n, nn = C.c_int64(0), C.c_int32(0)
        L.check(lib.natac_bedtab_dims(h, C.byref(n), C.byref(nn)))
        names = []
        for i in range(nn.value):
            buf = C.create_string_buffer(4096)
            L.check(lib.natac_bedtab_name(h, i, buf, 4096))
            names.append(buf.value.decode())
        cid, start, end = np.empty(n.value, np.int32), np.empty(n.value, np.int64), np.empty(n.value, np.int64)
        vals = np.empty((n.value, len(c)), np.float64)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        L.check(lib.natac_bedtab_fetch(h, vp(cid), vp(start), vp(end), vp(vals)))
    finally:
        lib.natac_bedtab_close(h)
    return names, cid, start, end, vals


def bgzip_file(src, dst=None, level=4, n_threads=0, remove=True):
    """BGZF-compress a text file (pysam.tabix_compress of the reference); returns the path of the .gz"""
    import os
    lib = L.load()
    dst = dst or (str(src) + ".gz")
    L.check(lib.natac_bgzip_file(str(src).encode(), str(dst).encode(), int(level), int(n_threads)))
    if remove:
        os.remove(src)
    return dst


def tabix_index(path, tbi_path=None, n_threads=0):
    """write path + '.tbi' ("bed" preset; pysam.tabix_index(path, preset="bed") of the reference); returns #records"""
    lib = L.load()
    n = C.c_int64(0)
    L.check(lib.natac_tabix_index(str(path).encode(), None if tbi_path is None else str(tbi_path).encode(), int(n_threads),
                                  C.byref(n)))
    return n.value


def bgzf_lines_host(text):
    """host restatement of the device BGZF encoder (natac_bgzf_lines_host): the members natac_batch_format_track(compress) produces
    for `text` (bytes, complete lines), without the EOF marker"""
    lib = L.load()
    text = bytes(text)
    a = np.frombuffer(text, dtype=np.uint8)
    off = np.concatenate(([0], np.flatnonzero(a[:-1] == 10) + 1)).astype(np.int64) if len(a) else np.zeros(0, np.int64)
    out = np.empty(len(text) + (len(text) // 0xff00 + 2) * 64 + 1024, dtype=np.uint8)
    nb = C.c_int64(0)
    L.check(lib.natac_bgzf_lines_host(text, len(text), off.ctypes.data_as(C.c_void_p), len(off), out.ctypes.data_as(C.c_void_p),
                                      out.nbytes, C.byref(nb)))
    return out[:nb.value].tobytes()


BGZF_EOF = bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])


class TbiBuilder(object):
    """incremental tabix index (natac_tbi) of a bedGraph.gz assembled from device-formatted results: push(index, file_offset) for
    every result in file order (`index` = info["index"] of DeviceBatch.format_track, `file_offset` = byte at which the result's
    first member sits in the file), then write(path + ".tbi") -- the file itself is never read again"""

    def __init__(self):
        self._lib = L.load()
        h = C.c_void_p()
        L.check(self._lib.natac_tbi_create(C.byref(h)))
        self._h = h

    def push(self, index, file_offset):
        n = len(index["cid"])
        if n == 0:
            return
        names = index["names"]
        arr = (C.c_char_p * len(names))(*[c.encode("ascii") for c in names])
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        L.check(self._lib.natac_tbi_push(self._h, n, arr, len(names), vp(index["cid"]), vp(index["beg"]), vp(index["end"]), vp(index["count"]),
                                         vp(index["t0"]), vp(index["t1"]), vp(index["member_pos"]), len(index["member_pos"]) - 1,
                                         int(index["n_text"]), int(file_offset)))

    def write(self, tbi_path):
        n = C.c_int64(0)
        L.check(self._lib.natac_tbi_write(self._h, str(tbi_path).encode(), C.byref(n)))
        return n.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.natac_tbi_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
