"""Genomic intervals: Chunk and ChunkList (API of the reference's pyatac/chunk.py:11-215).

Half-open [start, end) intervals on a chromosome; a ChunkList is a python list with BED reading, slop, merge
and split.  Integer arithmetic follows the reference's Python-2 floor division.
"""
import functools
import gzip
import warnings


class Chunk(object):
    """one genomic interval (pyatac/chunk.py:11-54)"""

    def __init__(self, chrom, start, end, weight=1, name="region", strand="*"):
        self.chrom = chrom
        self.start = start
        self.end = end
        self.weight = weight
        self.strand = strand
        self.name = name

    def length(self):
        return self.end - self.start

    def asBed(self):
        return "\t".join(str(x) for x in (self.chrom, self.start, self.end, self.weight, self.name, self.strand))

    def slop(self, chromDict, up=0, down=0, new=False):
        """extend by `up` upstream / `down` downstream (strand aware), clipped to the chromosome"""
        lo, hi = (down, up) if self.strand == "-" else (up, down)
        s = max(0, self.start - lo)
        e = min(chromDict[self.chrom], self.end + hi)
        if new:
            return Chunk(self.chrom, s, e, weight=self.weight, name=self.name, strand=self.strand)
        self.start, self.end = s, e

    def center(self, new=False):
        half = self.length() // 2
        if self.strand == "-":
            e = self.end - half
            s = e - 1
        else:
            s = self.start + half
            e = s + 1
        if new:
            return Chunk(self.chrom, s, e, weight=self.weight, name=self.name, strand=self.strand)
        self.start, self.end = s, e


def _chunkCompare(a, b):
    """order by chromosome name then start (the reference's comparator has a typo in its last branch,
    pyatac/chunk.py:66; equal starts compare equal here as there)"""
    if a.chrom != b.chrom:
        return -1 if a.chrom < b.chrom else 1
    if a.start < b.start:
        return -1
    return 0


class ChunkList(list):
    """list of Chunk (pyatac/chunk.py:71-215)"""

    def __init__(self, *args):
        list.__init__(self, args)

    def extend(self, *args):
        if len(args) != 1:
            raise ValueError("Wrong number of arguments")
        if not isinstance(args[0], ChunkList):
            raise ValueError("Expecting ChunkList")
        list.extend(self, args[0])

    def append(self, *args):
        if len(args) != 1:
            raise ValueError("Wrong number of arguments")
        if not isinstance(args[0], Chunk):
            raise ValueError("Expecting Chunk")
        list.append(self, args[0])

    def insert(self, *args):
        if len(args) != 2:
            raise ValueError("Wrong number of arguments")
        if not isinstance(args[1], Chunk):
            raise ValueError("Expecting Chunk")
        list.insert(self, args[0], args[1])

    def sort(self):
        list.sort(self, key=functools.cmp_to_key(_chunkCompare))

    def isSorted(self):
        # _chunkCompare(a, b) == -1 for every neighbouring pair, without a call per pair (10^5 regions per genome-wide run)
        prev = None
        for c in self:
            if prev is not None and not (prev.chrom < c.chrom or (prev.chrom == c.chrom and prev.start < c.start)):
                return False
            prev = c
        return True

    def slop(self, chromDict, up=0, down=0, new=False):
        out = ChunkList()
        add = list.append
        for c in self:               # Chunk.slop(new=True) inlined
            lo, hi = (down, up) if c.strand == "-" else (up, down)
            s, e = c.start - lo, c.end + hi
            n = chromDict[c.chrom]
            add(out, Chunk(c.chrom, s if s > 0 else 0, e if e < n else n, c.weight, c.name, c.strand))
        if new:
            return out
        self[:] = out

    def merge(self, new=False, sep=-1):
        """merge overlapping / nearby regions: next.start <= previous.end + sep joins (pyatac/chunk.py:109-125)"""
        if not self.isSorted():
            self.sort()
        out = ChunkList()
        if len(self) > 0:
            prev = self[0]
            add = list.append
            for cur in self[1:]:
                if cur.chrom == prev.chrom and cur.start <= prev.end + sep:
                    prev.end = max(cur.end, prev.end)
                else:
                    add(out, prev)
                    prev = cur
            add(out, prev)
        if new:
            return out
        self[:] = out

    def asBed(self):
        return "".join(c.asBed() + "\n" for c in self)

    @staticmethod
    def read(bedfile, weight_col=None, strand_col=None, name_col=None, chromDict=None, min_offset=None, min_length=1,
             chrom_source="FASTA file"):
        """read a (gzipped) BED file (pyatac/chunk.py:133-175); regions are clipped to
        [min_offset, chrom_len - min_offset] and dropped when shorter than min_length"""
        opener = gzip.open if bedfile[-3:] == ".gz" else open
        out = ChunkList()
        bad = []
        weight, strand, name = None, "+", None
        with opener(bedfile, "rt") as fh:
            for line in fh:
                f = line.rstrip("\n").split("\t")
                if len(f) < 3:
                    continue
                if weight_col:
                    weight = f[weight_col - 1]
                if strand_col:
                    strand = f[strand_col - 1]
                if name_col:
                    name = f[name_col - 1]
                chrom, start, end = f[0], int(f[1]), int(f[2])
                if chromDict is not None and chrom not in chromDict:
                    bad.append(chrom)
                    continue
                if min_offset:
                    start = max(start, min_offset)
                    end = min(end, chromDict[chrom] - min_offset)
                if end - start >= min_length:
                    list.append(out, Chunk(chrom, start, end, weight, name, strand))
        if bad:
            bad = sorted(set(bad))
            warnings.warn("%d chromosome names in bed file not included in %s:\n%s\n These regions will be ignored in "
                          "subsequent analysis" % (len(bad), chrom_source, "\n".join(bad)))
        return out

    @staticmethod
    def convertChromSizes(chromDict, splitsize=None, offset=0):
        out = ChunkList()
        for chrom in sorted(chromDict.keys()):
            if splitsize is None:
                out.append(Chunk(chrom, offset, chromDict[chrom] - offset))
            else:
                for i in range(offset, chromDict[chrom] - offset, splitsize):
                    out.append(Chunk(chrom, i, min(i + splitsize, chromDict[chrom] - offset)))
        return out

    def split(self, bases=None, items=None):
        """sub-lists of ~`bases` bases or exactly `items` chunks (pyatac/chunk.py:188-208)"""
        if bases is not None:
            out, i, acc, k = [], 0, 0, 0
            for k in range(len(self)):
                acc += self[k].length()
                if acc > bases:
                    out.append(self[i:k + 1])
                    acc = 0
                    i = k + 1
            if k >= i:
                out.append(self[i:k + 1])
            return out
        if items is not None:
            return [self[i:i + items] for i in range(0, len(self), items)]
        raise Exception("Need to provide items or bases argument!")

    def checkChroms(self, chroms, chunklist_source="bed file", chrom_source="fasta file",
                    warn="Regions on these chromosomes will be ignored in analysis"):
        bad = sorted(set(x.chrom for x in self if x.chrom not in chroms))
        if bad:
            self[:] = [x for x in self if x.chrom in chroms]
            warnings.warn("%d chromosome names in %s not included in %s:\n%s\n %s" % (
                len(bad), chunklist_source, chrom_source, "\n".join(bad), warn))
