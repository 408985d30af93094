"""V-plot template container (API of the reference's pyatac/VMat.py:21-218).

Only what the occ + nuc path needs runs here: the container, trim, and the text format.  Processing a raw
V-plot (`nucleoatac vprocess`: symmetrize / norm_y / smooth / norm) is a one-off host step outside the hot path;
`nucleoatac_amd/data/default_vmat.npz` ships the result for the default parameters.
"""
import os

import numpy as np

from .tracks import _py2_float_str


class VMat_Error(Exception):
    def __init__(self, value):
        self.value = value

    def __str__(self):
        return repr(self.value)


class VMat(object):
    def __init__(self, mat, lower, upper):
        if mat.shape[0] != upper - lower:
            raise VMat_Error("mat shape is not consistent with insert limits")
        self.mat = mat
        self.upper = upper
        self.lower = lower
        self.w = mat.shape[1] // 2

    def trim(self, lower, upper, w):
        up, dn = upper - self.lower, lower - self.lower
        left, right = self.w - w, self.w + w + 1
        if up > self.mat.shape[0] or dn < 0 or left < 0 or right > self.mat.shape[1]:
            raise VMat_Error("Mat is smaller than desired trim")
        self.mat = self.mat[dn:up, left:right]
        self.lower, self.upper, self.w = lower, upper, w

    def save(self, filename):
        with open(filename, "w") as out:
            out.write("#VMat Descriptor File\n#Contains VMat and pertinent information\n")
            out.write("#lower\n%d\n#upper\n%d\n#mat\n" % (self.lower, self.upper))
            for row in self.mat:
                out.write("\t".join(_py2_float_str(float(x)) for x in row) + "\n")

    @staticmethod
    def open(filename):
        if filename.endswith(".npz"):
            d = np.load(filename, allow_pickle=False)
            return VMat(np.array(d["vmat"]), int(d["vlower"]), int(d["vupper"]))
        state, lower, upper, rows = "", None, None, []
        with open(filename) as f:
            for line in f:
                if "#lower" in line:
                    state = "lower"
                elif "#upper" in line:
                    state = "upper"
                elif "#mat" in line:
                    state = "mat"
                elif "#" in line:
                    state = "other"
                elif state == "lower":
                    lower = int(line.strip())
                elif state == "upper":
                    upper = int(line.strip())
                elif state == "mat":
                    rows.append([float(x) for x in line.rstrip("\n").split("\t") if x != ""])
        if lower is None or upper is None or not rows:
            raise VMat_Error("VMat decriptor file appeas to be missing some needed components")
        return VMat(np.array(rows), lower, upper)

    @staticmethod
    def default():
        """the processed default V-plot (vprocess defaults: lower 105, upper 251, flank 60, smooth 0.75)"""
        return VMat.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "default_vmat.npz"))
