"""Native bedGraph writer (natac_write_bedgraph): all tracks of a batch in one multi-threaded call."""
import ctypes as C

import numpy as np

from . import _lib as L


def write_bedgraph(path, chroms, chunk_start, out_off, vals, append=False, compress=0, finish=True, write_zero=True,
                   n_threads=0):
    """run-length bedGraph text (Track.write_track semantics, pyatac/tracks.py:37-74) for many chunks at once.
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
                                     vals.ctypes.data_as(C.c_void_p), 1 if write_zero else 0, int(n_threads), C.byref(nb)))
    return nb.value
