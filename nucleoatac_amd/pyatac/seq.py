"""Sequence access (API of the reference's pyatac/seq.py:11-45) without pysam: FastaStore reads a (gzipped)
FASTA once, or an .npz of per-chromosome byte arrays."""
import gzip
import os

import numpy as np

_CACHE = {}
_PENDING = {}     # src -> (thread, result box) of FastaStore.prefetch


class FastaStore(object):
    def __init__(self, seqs):
        self.seqs = seqs                       # chrom -> np.uint8 array (upper-case ASCII)
        self.references = list(seqs.keys())
        self.lengths = [len(seqs[c]) for c in self.references]

    def chrom_sizes(self):
        return dict(zip(self.references, self.lengths))

    @staticmethod
    def sizes(src):
        """{record: length} without loading the sequence when that is possible: the .npy headers inside a .npz, the .fai next to a
        text FASTA (what pysam.FastaFile reads, pyatac/utils.py:33-40); otherwise the lengths of the loaded store"""
        if isinstance(src, FastaStore):
            return src.chrom_sizes()
        if src in _CACHE and src not in _PENDING:
            return _CACHE[src].chrom_sizes()
        try:
            if src.endswith(".npz"):
                import zipfile
                out = {}
                with zipfile.ZipFile(src) as z:
                    with z.open("chrom_names.npy") as fh:
                        names = [str(c) for c in np.lib.format.read_array(fh, allow_pickle=False)]
                    for c in names:
                        with z.open("seq_%s.npy" % c) as fh:
                            major, _minor = np.lib.format.read_magic(fh)
                            shape = (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(fh)[0]
                            out[c] = int(shape[0])
                return out
            if os.path.exists(src + ".fai"):
                out = {}
                with open(src + ".fai") as fh:
                    for line in fh:
                        f = line.split("\t")
                        if len(f) >= 2:
                            out[f[0]] = int(f[1])
                return out
        except Exception:      # noqa: BLE001 -- any surprise in the shortcut: the loaded store answers
            pass
        return FastaStore.open(src).chrom_sizes()

    @staticmethod
    def prefetch(src):
        """start loading `src` on a thread (native loader / zlib: the GIL is released); FastaStore.open(src) waits for it"""
        if isinstance(src, FastaStore) or src is None or src in _CACHE or src in _PENDING:
            return
        import threading
        box = {}

        def work():
            try:
                box["store"] = FastaStore._load(src)
            except BaseException as e:      # noqa: BLE001 -- re-raised by open() on the caller's thread
                box["error"] = e
        t = threading.Thread(target=work, name="natac-fasta-prefetch", daemon=True)
        _PENDING[src] = (t, box)
        t.start()

    @staticmethod
    def open(src):
        if isinstance(src, FastaStore):
            return src
        if src in _PENDING:
            t, box = _PENDING.pop(src)
            t.join()
            if "error" in box:
                raise box["error"]
            _CACHE[src] = box["store"]
        if src in _CACHE:
            return _CACHE[src]
        _CACHE[src] = st = FastaStore._load(src)
        return st

    @staticmethod
    def _load(src):
        if src.endswith(".npz"):
            d = np.load(src, allow_pickle=False)
            seqs = {}
            for c in d["chrom_names"]:
                a = d["seq_" + str(c)]
                if a.dtype != np.uint8 or (a >= 97).any():          # lower case somewhere: upper-case a copy
                    a = np.frombuffer(a.tobytes().upper(), dtype=np.uint8)
                seqs[str(c)] = a
            st = FastaStore(seqs)
        elif not src.endswith(".gz") and FastaStore._native_ok():
            st = FastaStore(FastaStore._load_native(src))
        else:
            opener = gzip.open if src.endswith(".gz") else open
            seqs, name, parts = {}, None, []
            with opener(src, "rt") as fh:
                for line in fh:
                    if line.startswith(">"):
                        if name is not None:
                            seqs[name] = np.frombuffer("".join(parts).upper().encode("ascii"), dtype=np.uint8)
                        name, parts = line[1:].split()[0], []
                    else:
                        parts.append(line.strip())
            if name is not None:
                seqs[name] = np.frombuffer("".join(parts).upper().encode("ascii"), dtype=np.uint8)
            st = FastaStore(seqs)
        return st

    @staticmethod
    def _native_ok():
        try:
            from .. import _lib as L
            L.load()
            return True
        except (ImportError, OSError, AttributeError):
            return False

    @staticmethod
    def _load_native(path):
        """plain-text FASTA through natac_fasta_* (csrc/natac_fasta.hpp: multi-threaded, memory speed) instead of a Python line
        loop over the whole genome"""
        import ctypes as C
        from .. import _lib as L
        lib = L.load()
        h = C.c_void_p()
        L.check(lib.natac_fasta_open(str(path).encode(), 0, C.byref(h)))
        try:
            n = C.c_int32(0)
            L.check(lib.natac_fasta_count(h, C.byref(n)))
            seqs = {}
            for r in range(n.value):
                name = C.create_string_buffer(4096)
                ln = C.c_int64(0)
                L.check(lib.natac_fasta_info(h, r, name, 4096, C.byref(ln)))
                a = np.empty(ln.value, dtype=np.uint8)
                L.check(lib.natac_fasta_read(h, r, a.ctypes.data_as(C.c_void_p), ln.value))
                seqs[name.value.decode()] = a
        finally:
            lib.natac_fasta_close(h)
        return seqs

    def fetch(self, chrom, start, end):
        return self.seqs[chrom][start:end].tobytes().decode("ascii")


_COMPLEMENT = str.maketrans("ACGT", "TGCA")


def complement(sequence):
    return sequence.translate(_COMPLEMENT)


def reverse_complement(sequence):
    return complement(sequence[::-1])


def get_sequence(chunk, fastafile):
    """upper-case sequence of an interval, reverse-complemented on the minus strand (pyatac/seq.py:11-22)"""
    s = FastaStore.open(fastafile).fetch(chunk.chrom, chunk.start, chunk.end)
    if chunk.strand == "-":
        s = reverse_complement(s)
    return s.upper()


def seq_to_mat(sequence, nucleotides):
    """one-hot encoding, one row per (equal-length) nucleotide word (pyatac/seq.py:37-45)"""
    k = len(nucleotides[0])
    if not all(len(x) == k for x in nucleotides):
        raise Exception("Usage Error! Nucleotides must all be of same length! No mixing single nucleotides with dinucleotides, etc")
    n = len(sequence) - k + 1
    mat = np.zeros((len(nucleotides), n))
    for i, word in enumerate(nucleotides):
        mat[i] = [1.0 if sequence[j:j + k] == word else 0.0 for j in range(n)]
    return mat
