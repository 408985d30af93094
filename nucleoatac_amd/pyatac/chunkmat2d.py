"""2-D (insert size x position) matrices of a chunk (API of the reference's pyatac/chunkmat2d.py:9-156).

The batched GPU path never materialises these matrices; the classes exist for the operator-level API
(single-chunk use, tests written like the reference's) and build their matrices on the GPU.
"""
import numpy as np

from .tracks import InsertionTrack


class ChunkMat2D(object):
    """rows = insert sizes [lower, upper), columns = positions [start, end)"""

    def __init__(self, chrom, start, end, lower, upper):
        self.chrom = chrom
        self.lower = lower
        self.upper = upper
        self.start = start
        self.end = end
        self.ncol = end - start
        self.nrow = upper - lower
        self.mat = np.zeros((self.nrow, self.ncol))

    def get(self, lower=None, upper=None, start=None, end=None, flip=False):
        """sub-matrix by insert size / genomic position (pyatac/chunkmat2d.py:21-54)"""
        lower = self.lower if lower is None else lower
        upper = self.upper if upper is None else upper
        start = self.start if start is None else start
        end = self.end if end is None else end
        y1, y2 = lower - self.lower, upper - self.lower
        x1, x2 = start - self.start, end - self.start
        if not flip:
            try:
                return self.mat[y1:y2, x1:x2]
            except Exception:
                raise Exception("Looks like dimensions from get probaby don't match Mat")
        if x1 < 1 or x2 > self.mat.shape[1] or y1 < 0 or y2 > self.mat.shape[0]:
            raise Exception("Looks like dimensions from get probaby don't match Mat")
        if (x2 - x1) % 2 == 0:
            raise Exception("Can only flip mat if the width is odd!")
        new = np.zeros((y2 - y1, x2 - x1))
        for j in range(y1, y2):
            if (j + self.lower) % 2 == 1:
                new[j - y1, :] = self.mat[j, x1:x2][::-1]
            else:
                new[j - y1, :] = self.mat[j, (x1 - 1):x2][::-1][1:]
        return new

    def assign(self, mat):
        if mat.shape != self.mat.shape:
            raise Exception("Dimensions of input mat are wrong.  Uh oh!")
        self.mat = mat

    def save(self, filename):
        head = ",".join(str(x) for x in (self.chrom, self.start, self.end, self.lower, self.upper))
        np.savetxt(filename, self.mat, delimiter="\t", header=head)

    @staticmethod
    def open(filename):
        with open(filename) as f:
            el = f.readline().rstrip("\n").lstrip("#").strip().split(",")
        new = ChunkMat2D(el[0], int(el[1]), int(el[2]), int(el[3]), int(el[4]))
        new.assign(np.loadtxt(filename, skiprows=1))
        return new

    def getIns(self):
        """collapse the matrix to insertions (pyatac/chunkmat2d.py:74-84): a fragment of size i centred at column x
        inserts at x - (i-1)//2 and x + i//2 (a single position when the two coincide, i == 1).  The output spans
        [start + P//2, end - P//2), P = upper + (upper-1) % 2."""
        P = self.upper + (self.upper - 1) % 2
        half = P // 2
        nout = self.mat.shape[1] - P + 1
        ins = np.zeros(nout)
        rows, cols = np.nonzero(self.mat)
        for r, c in zip(rows, cols):
            i = r + self.lower
            for t in {c - (i - 1) // 2, c + i // 2}:
                o = t - half
                if 0 <= o < nout:
                    ins[o] += self.mat[r, c]
        track = InsertionTrack(self.chrom, self.start + half, self.end - half)
        track.assign_track(ins)
        return track


class FragmentMat2D(ChunkMat2D):
    """fragment-centre counts (pyatac/chunkmat2d.py:112-132)"""

    def __init__(self, chrom, start, end, lower, upper, atac=True):
        ChunkMat2D.__init__(self, chrom, start, end, lower, upper)
        self.atac = atac

    def makeFragmentMat(self, bamfile):
        from .fragments import makeFragmentMat
        self.mat = makeFragmentMat(bamfile, self.chrom, self.start, self.end, self.lower, self.upper, self.atac)


class BiasMat2D(ChunkMat2D):
    """expected relative fragment frequencies from the Tn5 bias model (pyatac/chunkmat2d.py:135-156)"""

    def __init__(self, chrom, start, end, lower, upper):
        ChunkMat2D.__init__(self, chrom, start, end, lower, upper)
        self.mat = np.ones(self.mat.shape)

    def makeBiasMat(self, bias_track):
        from .. import get_context
        offset = self.upper // 2
        bias = bias_track.get(self.start - offset, self.end + offset)
        if len(bias) != self.ncol + 2 * offset:
            raise Exception("Looks like dimensions from get probaby don't match track, or there are no vals in track")
        if not bias_track.log:
            nz = bias[bias != 0]
            bias = np.log(bias + np.min(nz))
        self.mat = get_context().make_bias_mat(bias, self.start - offset, self.start, self.end, self.lower, self.upper)

    def normByInsertDist(self, insertsizes):
        inserts = np.asarray(insertsizes.get(self.lower, self.upper), dtype=np.float64)
        self.mat = self.mat * inserts[:, None]
