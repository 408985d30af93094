"""Sequence access (API of the reference's pyatac/seq.py:11-45) without pysam: FastaStore reads a (gzipped)
FASTA once, or an .npz of per-chromosome byte arrays."""
import gzip

import numpy as np

_CACHE = {}


class FastaStore(object):
    def __init__(self, seqs):
        self.seqs = seqs                       # chrom -> np.uint8 array (upper-case ASCII)
        self.references = list(seqs.keys())
        self.lengths = [len(seqs[c]) for c in self.references]

    def chrom_sizes(self):
        return dict(zip(self.references, self.lengths))

    @staticmethod
    def open(src):
        if isinstance(src, FastaStore):
            return src
        if src in _CACHE:
            return _CACHE[src]
        if src.endswith(".npz"):
            d = np.load(src, allow_pickle=False)
            st = FastaStore({str(c): np.frombuffer(d["seq_" + str(c)].tobytes().upper(), dtype=np.uint8)
                             for c in d["chrom_names"]})
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
        _CACHE[src] = st
        return st

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
