"""`nucleoatac merge` (reference: nucleoatac/merge.py): combine the occupancy peaks of `occ` with the nucleosome calls of
`nuc` into one position map.  Pure list logic on two sorted BED files (host); output bgzip + tabix like every other output."""
import gzip
import os

from ..pyatac.chunk import Chunk, ChunkList
from ..pyatac.tracks import _py2_float_str
from ..writer import bgzip_file, tabix_index


def _num(x):
    """py2 `str()` of a parsed column: ints stay ints, floats get the 12-digit form the reference writes"""
    return str(x) if isinstance(x, int) else _py2_float_str(float(x))


class MergedNuc(Chunk):
    """one row of the combined map (merge.py:15-30)"""

    def __init__(self, chrom, start, end, occ, occ_lower, occ_upper, reads, source):
        self.chrom = chrom
        self.start = start
        self.end = end
        self.occ = occ
        self.occ_lower = occ_lower
        self.occ_upper = occ_upper
        self.reads = reads
        self.source = source

    def asBed(self):
        return "\t".join([str(self.chrom), str(self.start), str(self.end), _num(self.occ), _num(self.occ_lower),
                          _num(self.occ_upper), _num(self.reads), str(self.source)])

    def write(self, handle):
        handle.write(self.asBed() + "\n")


class NucList(ChunkList):
    def __init__(self, *args):
        list.__init__(self, args)

    @staticmethod
    def read(bedfile, source, min_occ=0):
        """rows of an occpeaks.bed (source 'occ') or nucpos.bed (source 'nuc') file with occ_lower >= min_occ
        (merge.py:36-64; NaN lower bounds fail the comparison and are dropped like in the reference)"""
        if source not in ("occ", "nuc"):
            raise Exception("source must be 'occ' or 'nuc'")
        opener = gzip.open if bedfile[-3:] == ".gz" else open
        out = NucList()
        with opener(bedfile, "rt") as infile:
            for line in infile:
                f = line.rstrip("\n").split("\t")
                start, end = int(f[1]), int(f[2])
                if source == "occ":
                    occ, lo, up, reads = float(f[3]), float(f[4]), float(f[5]), float(f[6])
                else:
                    occ, lo, up, reads = float(f[4]), float(f[5]), float(f[6]), float(f[10]) + float(f[11])
                if lo >= min_occ:
                    out.append(MergedNuc(f[0], start, end, occ, lo, up, reads, source))
        return out


def merge(occ_peaks, nuc_calls, sep=120):
    """two-pointer merge (merge.py:69-96): nucleosome calls win; an occupancy peak is kept only if no call lies within
    `sep` of it (chromosomes compared as strings, both lists sorted)"""
    keep = NucList()
    i = j = 0
    while i < len(occ_peaks) and j < len(nuc_calls):
        if occ_peaks[i].chrom < nuc_calls[j].chrom:
            keep.append(occ_peaks[i])
            i += 1
        elif occ_peaks[i].chrom > nuc_calls[j].chrom:
            keep.append(nuc_calls[j])
            j += 1
        elif occ_peaks[i].start < (nuc_calls[j].start - sep):
            keep.append(occ_peaks[i])
            i += 1
        elif occ_peaks[i].start > (nuc_calls[j].start + sep):
            keep.append(nuc_calls[j])
            j += 1
        else:
            i += 1
    keep.extend(NucList(*nuc_calls[j:]))
    keep.extend(NucList(*occ_peaks[i:]))
    return keep


def run_merge(args):
    if not args.out:
        args.out = ".".join(os.path.basename(args.nucpos).split(".")[0:-3])
    occ = NucList.read(args.occpeaks, "occ", float(args.min_occ))
    nuc = NucList.read(args.nucpos, "nuc", float(args.min_occ))
    new = merge(occ, nuc, int(args.sep))
    path = args.out + ".nucmap_combined.bed"
    with open(path, "w") as out:
        out.write(new.asBed())
    bgzip_file(path)                       # pysam.tabix_compress + rm + tabix_index (merge.py:107-109)
    tabix_index(path + ".gz")
    return new
