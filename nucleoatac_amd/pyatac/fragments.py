"""Fragment access + the four extractor functions of the reference's Cython module pyatac/fragments.pyx.

The reference re-opens the BAM and iterates `AlignmentFile.fetch` inside every call (fragments.pyx:21-25).
Here the alignment file is decoded ONCE into a FragmentStore (per chromosome: sorted leftmost positions and
template lengths of the forward proper-pair reads, which are the only reads the reference keeps,
fragments.pyx:25); the extractor functions keep the reference's names and argument order -- `bamfile` may be a
FragmentStore or a path (.bam / .npz) -- and run on the GPU through the C-ABI.
"""
import gzip
import os
import struct

import numpy as np

_CACHE = {}
_PENDING = {}     # src -> (thread, result box) of FragmentStore.prefetch


class FragmentStore(object):
    """per-chromosome arrays of forward proper-pair reads: pos (leftmost coordinate, sorted), tlen (|template length|)"""

    def __init__(self, chroms, lengths, pos, tlen, trusted=False):
        self.references = list(chroms)
        self.lengths = [int(x) for x in lengths]
        if trusted:      # arrays published by another FragmentStore (shard.shared_fragment_store): already int64, |tlen|, sorted
            self.pos = {c: pos[c] for c in self.references}
            self.tlen = {c: tlen[c] for c in self.references}
            self.max_tlen = max([int(t.max()) for t in self.tlen.values() if len(t)] + [0])
            return
        self.pos = {c: np.ascontiguousarray(pos[c], dtype=np.int64) for c in self.references}
        self.tlen = {c: np.ascontiguousarray(np.abs(tlen[c]), dtype=np.int64) for c in self.references}
        for c in self.references:
            if len(self.pos[c]) > 1 and np.any(np.diff(self.pos[c]) < 0):
                o = np.argsort(self.pos[c], kind="stable")
                self.pos[c], self.tlen[c] = self.pos[c][o], self.tlen[c][o]
        self.max_tlen = max([int(t.max()) for t in self.tlen.values() if len(t)] + [0])

    def chrom_sizes(self):
        return dict(zip(self.references, self.lengths))

    @staticmethod
    def prefetch(src):
        """start decoding `src` on a thread (the drivers call this first thing: the BAM decode -- native code, GIL released --
        runs while the main thread reads the FASTA index and the BED file); FragmentStore.open(src) waits for it"""
        if isinstance(src, FragmentStore) or src in _CACHE or src in _PENDING:
            return
        import threading
        if os.environ.get("NATAC_DEVICE_BAM", "1") != "0" and str(src).endswith(".bam"):
            from .. import get_context
            from ..device import Context
            if Context.device_count() > 0:
                get_context()               # the process-wide context is created on the caller's thread, not raced for
        box = {}

        def work():
            try:
                box["store"] = FragmentStore._load(src)
            except BaseException as e:      # noqa: BLE001 -- re-raised by open() on the caller's thread
                box["error"] = e
        t = threading.Thread(target=work, name="natac-bam-prefetch", daemon=True)
        _PENDING[src] = (t, box)
        t.start()

    @staticmethod
    def open(src):
        if isinstance(src, FragmentStore):
            return src
        if src in _PENDING:
            t, box = _PENDING.pop(src)
            t.join()
            if "error" in box:
                raise box["error"]
            _CACHE[src] = box["store"]
        if src in _CACHE:
            return _CACHE[src]
        _CACHE[src] = st = FragmentStore._load(src)
        return st

    @staticmethod
    def _load(src):
        if src.endswith(".npz"):
            st = FragmentStore.from_npz(src)
        elif src.endswith(".bam"):
            st = FragmentStore.from_bam(src)
        else:
            raise ValueError("unsupported alignment source %r (expected FragmentStore, .bam or .npz)" % (src,))
        return st

    @staticmethod
    def register(src, store):
        """make FragmentStore.open(src) return `store` (a store built elsewhere, e.g. mapped from shared memory)"""
        _CACHE[src] = store

    @staticmethod
    def from_npz(path):
        d = np.load(path, allow_pickle=False)
        chroms = [str(x) for x in d["chrom_names"]]
        return FragmentStore(chroms, d["chrom_lengths"], {c: d["pos_" + c] for c in chroms},
                             {c: d["tlen_" + c] for c in chroms})

    def save_npz(self, path):
        arrs = dict(chrom_names=np.array(self.references), chrom_lengths=np.array(self.lengths))
        for c in self.references:
            arrs["pos_" + c] = self.pos[c]
            arrs["tlen_" + c] = self.tlen[c]
        np.savez_compressed(path, **arrs)

    @staticmethod
    def from_bam(path, n_threads=0, device=None):
        """native decoder in libnatac_hip.so.  On a GPU box: natac_bam_open_device (csrc/natac_bam_dev.hpp) -- the file goes to the
        device through two pinned staging buffers, every lane of the chip inflates BGZF members from a queue while the rest of
        the file is still being read, the records are walked on the device and the chain of record starts is confirmed link by
        link.  Without a GPU, or with device=False / NATAC_DEVICE_BAM=0: natac_bam_open, parallel inflate + record walk on the
        host cores.  Both give the same arrays.  Measured on the MI355X box (tools/bench_bam.py, 60 M records, 4.6 GB): device
        0.72 s; host 1.64 s with its 64 threads, 6.1 s with 4."""
        import ctypes as C
        from .. import _lib as L
        lib = L.load()
        h = C.c_void_p()
        if device is None:
            from ..device import Context
            device = os.environ.get("NATAC_DEVICE_BAM", "1") != "0" and Context.device_count() > 0
        if device:
            from .. import get_context
            on_dev = C.c_int(0)
            L.check(lib.natac_bam_open_device(get_context()._h, str(path).encode(), C.byref(h), C.byref(on_dev)))
            FragmentStore.last_bam_on_device = bool(on_dev.value)
        else:
            L.check(lib.natac_bam_open(str(path).encode(), int(n_threads), C.byref(h)))
            FragmentStore.last_bam_on_device = False
        try:
            nref = C.c_int32(0)
            L.check(lib.natac_bam_counts(h, C.byref(nref), None, None))
            names, lens, pos, tl = [], [], {}, {}
            for r in range(nref.value):
                name = C.create_string_buffer(512)
                ln, nr = C.c_int64(0), C.c_int64(0)
                L.check(lib.natac_bam_ref_info(h, r, name, 512, C.byref(ln), C.byref(nr)))
                c = name.value.decode()
                p = np.empty(nr.value, dtype=np.int64)
                t = np.empty(nr.value, dtype=np.int64)
                L.check(lib.natac_bam_ref_reads(h, r, p.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), nr.value))
                names.append(c)
                lens.append(ln.value)
                pos[c], tl[c] = p, t
        finally:
            lib.natac_bam_close(h)
        return FragmentStore(names, lens, pos, tl)

    @staticmethod
    def from_bam_python(path):
        """pure-Python BGZF/BAM decoder (SAM spec section 4), kept as an independent check of the native one:
        keeps FLAG & 0x2 (proper pair) and not FLAG & 0x10 (reverse)"""
        with gzip.open(path, "rb") as fh:
            b = fh.read()
        if b[:4] != b"BAM\x01":
            raise ValueError("%s is not a BAM file" % path)
        o = 8 + struct.unpack_from("<i", b, 4)[0]
        n_ref = struct.unpack_from("<i", b, o)[0]
        o += 4
        names, lens = [], []
        for _ in range(n_ref):
            ln = struct.unpack_from("<i", b, o)[0]
            names.append(b[o + 4:o + 3 + ln].decode())
            lens.append(struct.unpack_from("<i", b, o + 4 + ln)[0])
            o += 8 + ln
        pos = {c: [] for c in names}
        tl = {c: [] for c in names}
        nb = len(b)
        while o + 36 <= nb:
            bs, ref_id, p = struct.unpack_from("<iii", b, o)
            flag = struct.unpack_from("<H", b, o + 18)[0]
            tlen = struct.unpack_from("<i", b, o + 32)[0]
            if ref_id >= 0 and (flag & 0x2) and not (flag & 0x10):
                pos[names[ref_id]].append(p)
                tl[names[ref_id]].append(tlen)
            o += 4 + bs
        return FragmentStore(names, lens, {c: np.array(pos[c], np.int64) for c in names},
                             {c: np.array(tl[c], np.int64) for c in names})

    @staticmethod
    def from_arrays(chrom_sizes, l, n, atac=True):
        """build from already shifted fragments (l = pos+4, n = |tlen|-8 when atac)"""
        sh, d = (4, 8) if atac else (0, 0)
        chroms = list(chrom_sizes.keys())
        return FragmentStore(chroms, [chrom_sizes[c] for c in chroms],
                             {c: np.asarray(l.get(c, []), np.int64) - sh for c in chroms},
                             {c: np.asarray(n.get(c, []), np.int64) + d for c in chroms})

    def fetch(self, chrom, start, end, atac=1):
        """(l, n) of every read that can matter for [start, end): superset of htslib's overlap fetch
        (fragments.pyx:24); l = pos+4, n = |tlen|-8 when atac (fragments.pyx:26-34)"""
        if chrom not in self.pos:
            return np.zeros(0, np.int64), np.zeros(0, np.int32)
        p = self.pos[chrom]
        a = int(np.searchsorted(p, start - 1024, "left"))
        b = int(np.searchsorted(p, end, "left"))
        if atac:
            return p[a:b] + 4, (self.tlen[chrom][a:b] - 8).astype(np.int32)
        return p[a:b].copy(), self.tlen[chrom][a:b].astype(np.int32)

    def all_fragments(self, chrom, atac=1):
        return self.fetch(chrom, -(1 << 40), 1 << 40, atac)


def _ctx():
    from .. import get_context
    return get_context()


def makeFragmentMat(bamfile, chrom, start, end, lower, upper, atac=1):
    """V-plot count matrix (upper-lower) x (end-start) -- pyatac/fragments.pyx:17-40"""
    l, n = FragmentStore.open(bamfile).fetch(chrom, max(0, start - upper), end + upper, atac)
    return _ctx().make_fragment_mat(l, n, start, end, lower, upper)


def getInsertions(bamfile, chrom, start, end, lower, upper, atac=1):
    """per-base Tn5 insertion counts -- pyatac/fragments.pyx:43-67"""
    l, n = FragmentStore.open(bamfile).fetch(chrom, max(0, start - upper), end + upper, atac)
    return _ctx().get_insertions(l, n, start, end, lower, upper)


def getStrandedInsertions(bamfile, chrom, start, end, lower, upper, atac=1):
    """(plus, minus) = insertions at the left / right fragment ends -- pyatac/fragments.pyx:71-97"""
    l, n = FragmentStore.open(bamfile).fetch(chrom, max(0, start - upper), end + upper, atac)
    return _ctx().get_stranded_insertions(l, n, start, end, lower, upper)


def getAllFragmentSizes(bamfile, lower, upper, atac=1):
    """insert-size histogram of the whole file -- pyatac/fragments.pyx:101-119"""
    st = FragmentStore.open(bamfile)
    sizes = np.zeros(upper - lower, dtype=np.float64)
    for c, ln in zip(st.references, st.lengths):
        l, n = st.all_fragments(c, atac)
        if len(l):
            big = 1 << 40
            sizes += _ctx().fragment_sizes(l, n, [-big], [big], lower, upper)
    return sizes


def getFragmentSizesFromChunkList(chunks, bamfile, lower, upper, atac=1):
    """insert-size histogram of fragments centred in the chunks -- pyatac/fragments.pyx:123-145
    (a fragment counts once per chunk that contains its centre)"""
    st = FragmentStore.open(bamfile)
    sizes = np.zeros(upper - lower, dtype=np.float64)
    by_chrom = {}
    for ch in chunks:
        by_chrom.setdefault(ch.chrom, []).append((ch.start, ch.end))
    for c, iv in by_chrom.items():
        iv.sort()
        lo, hi = iv[0][0], max(e for _, e in iv)
        l, n = st.fetch(c, max(0, lo - upper), hi + upper, atac)
        if len(l):
            sizes += _ctx().fragment_sizes(l, n, [s for s, _ in iv], [e for _, e in iv], lower, upper)
    return sizes
