"""`nucleoatac merge` (reference: nucleoatac/merge.py): combine the occupancy peaks of `occ` with the nucleosome calls of
`nuc` into one position map.  Pure list logic on two sorted BED files (host); output bgzip + tabix like every other output."""
import gzip
import os

import numpy as np

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
        out = NucList()
        add = list.append                # (ChunkList.append validates every element: 10^5-10^6 rows per genome)
        try:                             # parsed natively: columns instead of 10^5-10^6 split() / float() rounds
            from ..writer import read_bed_table
            names, cid, start, end, v = read_bed_table(bedfile, (3, 4, 5, 6) if source == "occ" else (4, 5, 6, 10, 11))
        except ImportError:
            names = None
        if names is not None:
            if source == "nuc":
                v = np.column_stack([v[:, 0], v[:, 1], v[:, 2], v[:, 3] + v[:, 4]])
            with np.errstate(invalid="ignore"):
                keep = v[:, 1] >= min_occ                       # NaN lower bounds fail the comparison
            chrom = [names[i] for i in cid[keep].tolist()]
            for c, s, e, (o, lo, up, rd) in zip(chrom, start[keep].tolist(), end[keep].tolist(), v[keep].tolist()):
                add(out, MergedNuc(c, s, e, o, lo, up, rd, source))
            return out
        opener = gzip.open if bedfile[-3:] == ".gz" else open
        with opener(bedfile, "rt") as infile:
            lines = infile.read().split("\n")
        if source == "occ":
            for line in lines:
                if line:
                    f = line.split("\t")
                    lo = float(f[4])
                    if lo >= min_occ:
                        add(out, MergedNuc(f[0], int(f[1]), int(f[2]), float(f[3]), lo, float(f[5]), float(f[6]), source))
        else:
            for line in lines:
                if line:
                    f = line.split("\t")
                    lo = float(f[5])
                    if lo >= min_occ:
                        add(out, MergedNuc(f[0], int(f[1]), int(f[2]), float(f[4]), lo, float(f[6]), float(f[10]) + float(f[11]), source))
        return out


def merge(occ_peaks, nuc_calls, sep=120):
    """Combined map of two position-sorted lists (merge.py:69-96).  Nucleosome calls always survive; an occupancy peak
    survives only if the next unconsumed call on its chromosome is more than `sep` away from it (then whichever of the two
    comes first in (chromosome string, position) order is emitted).  A peak within `sep` of the pending call is dropped
    without consuming the call."""
    keep = NucList()
    add = list.append
    calls = iter(nuc_calls)
    call = next(calls, None)
    for peak in occ_peaks:
        while call is not None:
            if peak.chrom != call.chrom:
                ahead = call.chrom < peak.chrom            # the call's chromosome sorts first: emit it, look at the next call
            else:
                gap = peak.start - call.start
                if -sep <= gap <= sep:
                    peak = None                            # shadowed by the pending call
                    break
                ahead = gap > sep
            if not ahead:
                break
            add(keep, call)
            call = next(calls, None)
        if peak is not None:
            add(keep, peak)
    while call is not None:
        add(keep, call)
        call = next(calls, None)
    return keep


def write_rows(path, rows):
    """MergedNuc.asBed of every row (merge.py:23-30), formatted natively (natac_write_bed_rows_labeled: python-2 float text) when
    all four value columns are floats -- what NucList.read produces --, else row by row"""
    import numpy as np
    from ..writer import write_bed_rows
    if not len(rows) or not all(type(v) is float for r in rows for v in (r.occ, r.occ_lower, r.occ_upper, r.reads)):
        with open(path, "w") as out:
            out.write(rows.asBed() if isinstance(rows, ChunkList) else "".join(r.asBed() + "\n" for r in rows))
        return
    names = sorted(set(r.chrom for r in rows))
    idx = {c: i for i, c in enumerate(names)}
    labels = sorted(set(str(r.source) for r in rows))
    lidx = {c: i for i, c in enumerate(labels)}
    write_bed_rows(path, names, np.array([idx[r.chrom] for r in rows], dtype=np.int32), np.array([r.start for r in rows], dtype=np.int64),
                   np.array([r.end for r in rows], dtype=np.int64),
                   np.array([(r.occ, r.occ_lower, r.occ_upper, r.reads) for r in rows], dtype=np.float64).reshape(-1, 4), append=False,
                   labels=labels, label_id=np.array([lidx[str(r.source)] for r in rows], dtype=np.int32))


def _table(bedfile, source, min_occ):
    """(chromosome per row, start, end, [occ, occ_lower, occ_upper, reads]) of the rows NucList.read keeps, as columns"""
    from ..writer import read_bed_table
    names, cid, start, end, v = read_bed_table(bedfile, (3, 4, 5, 6) if source == "occ" else (4, 5, 6, 10, 11))
    if source == "nuc":
        v = np.column_stack([v[:, 0], v[:, 1], v[:, 2], v[:, 3] + v[:, 4]])
    with np.errstate(invalid="ignore"):
        keep = v[:, 1] >= min_occ
    return [names[i] for i in cid[keep].tolist()], start[keep], end[keep], v[keep]


def merge_columns(occ, nuc, sep=120):
    """`merge` on columns: (source, row) of every surviving row in output order, from the chromosome and start columns alone --
    the same walk as merge() without an object per row"""
    pc, ps = occ[0], occ[1].tolist()
    cc, cs = nuc[0], nuc[1].tolist()
    out_src, out_row = [], []
    j, nj = 0, len(cs)
    for i in range(len(ps)):
        chrom, x = pc[i], ps[i]
        shadowed = False
        while j < nj:
            if chrom != cc[j]:
                ahead = cc[j] < chrom
            else:
                gap = x - cs[j]
                if -sep <= gap <= sep:
                    shadowed = True
                    break
                ahead = gap > sep
            if not ahead:
                break
            out_src.append(1)
            out_row.append(j)
            j += 1
        if not shadowed:
            out_src.append(0)
            out_row.append(i)
    out_src += [1] * (nj - j)
    out_row += list(range(j, nj))
    return np.array(out_src, dtype=np.int32), np.array(out_row, dtype=np.int64)


def run_merge(args):
    if not args.out:
        args.out = ".".join(os.path.basename(args.nucpos).split(".")[0:-3])
    path = args.out + ".nucmap_combined.bed"
    try:
        occ_t = _table(args.occpeaks, "occ", float(args.min_occ))
        nuc_t = _table(args.nucpos, "nuc", float(args.min_occ))
    except ImportError:
        occ_t = None
    if occ_t is not None:
        # columns in, columns out: no object per row (10^5-10^6 rows per genome); the rows are MergedNuc.asBed's
        from ..writer import write_bed_rows
        src, row = merge_columns(occ_t, nuc_t, int(args.sep))
        is_occ = src == 0
        chrom = np.empty(len(src), dtype=object)
        chrom[is_occ] = np.array(occ_t[0], dtype=object)[row[is_occ]] if is_occ.any() else []
        chrom[~is_occ] = np.array(nuc_t[0], dtype=object)[row[~is_occ]] if (~is_occ).any() else []
        names = sorted(set(chrom.tolist()))
        idx = {c: i for i, c in enumerate(names)}
        start, end, vals = np.empty(len(src), np.int64), np.empty(len(src), np.int64), np.empty((len(src), 4))
        for m, t in ((is_occ, occ_t), (~is_occ, nuc_t)):
            start[m], end[m], vals[m] = t[1][row[m]], t[2][row[m]], t[3][row[m]]
        write_bed_rows(path, names, np.array([idx[c] for c in chrom.tolist()], dtype=np.int32), start, end, vals, append=False,
                       labels=["occ", "nuc"], label_id=src)
        bgzip_file(path)
        tabix_index(path + ".gz")
        return None
    occ = NucList.read(args.occpeaks, "occ", float(args.min_occ))
    nuc = NucList.read(args.nucpos, "nuc", float(args.min_occ))
    new = merge(occ, nuc, int(args.sep))
    path = args.out + ".nucmap_combined.bed"
    write_rows(path, new)
    bgzip_file(path)                       # pysam.tabix_compress + rm + tabix_index (merge.py:107-109)
    tabix_index(path + ".gz")
    return None                            # like the column path above and the reference's run_merge (merge.py:94-109): files only
