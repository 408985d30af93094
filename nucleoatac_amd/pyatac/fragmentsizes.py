"""Insert-size distribution (API of the reference's pyatac/fragmentsizes.py:14-84)."""
import numpy as np

from .tracks import _py2_float_str


class FragmentSizes(object):
    def __init__(self, lower, upper, atac=True, vals=None):
        self.lower = lower
        self.upper = upper
        self.vals = vals
        self.atac = atac

    def calculateSizes(self, bamfile, chunks=None):
        from .fragments import getAllFragmentSizes, getFragmentSizesFromChunkList
        if chunks is None:
            sizes = getAllFragmentSizes(bamfile, self.lower, self.upper, atac=self.atac)
        else:
            sizes = getFragmentSizesFromChunkList(chunks, bamfile, self.lower, self.upper, atac=self.atac)
        tot = np.sum(sizes)
        self.vals = sizes / (tot + (tot == 0))

    def get(self, lower=None, upper=None, size=None):
        if size:
            try:
                return self.vals[size - self.lower]
            except Exception:
                raise Exception("Looks like size doesn't match FragmentSizes")
        lower = self.lower if lower is None else lower
        upper = self.upper if upper is None else upper
        try:
            return self.vals[lower - self.lower:upper - self.lower]
        except Exception:
            raise Exception("Looks like dimensions from get probaby don't match FragmentSizes")

    def save(self, filename):
        with open(filename, "w") as f:
            f.write("#lower\n%d\n#upper\n%d\n#sizes\n" % (self.lower, self.upper))
            f.write("\t".join(_py2_float_str(float(x)) for x in self.get()) + "\n")

    @staticmethod
    def open(filename):
        state, vals = "", {}
        with open(filename) as f:
            for line in f:
                if line.startswith("#"):
                    key = line.strip().lstrip("#")
                    state = key if key in ("lower", "upper", "sizes") else "other"
                elif state in ("lower", "upper"):
                    vals[state] = int(line.strip())
                elif state == "sizes":
                    vals["sizes"] = np.array([float(x) for x in line.rstrip("\n").split("\t")])
        if not all(k in vals for k in ("lower", "upper", "sizes")):
            raise Exception("FragmentDistribution decriptor file appeas to be missing some needed components")
        return FragmentSizes(vals["lower"], vals["upper"], vals=vals["sizes"])
