"""V-plot template container (API of the reference's pyatac/VMat.py:21-218).

The container, the text format and the `nucleoatac vprocess` processing steps (trim / symmetrize / norm_y / smooth /
norm, nucleoatac/run_vprocess.py:20-33).  Processing a raw V-plot is a one-off host step on a 146 x 121 matrix outside
the hot path, so it stays numpy / scipy.ndimage like the reference; `nucleoatac_amd/data/standard_vplot.npz` holds the
reference's raw S. cer. V-plot (its data file nucleoatac/vplot/standard_vplot.VMat) and `default_vmat.npz` the result
of the default parameters with the example's nucleosomal size distribution.
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

    def symmetrize(self):
        """force the V-plot to be symmetric around the dyad (pyatac/VMat.py:54-63): odd insert sizes mirror around the
        centre column, even ones around the boundary left of it (their last column is kept)"""
        w = self.w
        for j in range(self.lower, self.upper):
            i = j - self.lower
            if j % 2 == 1:
                lefthalf = (self.mat[i, :(w + 1)] + self.mat[i, w:][::-1]) * 0.5
                self.mat[i, :] = np.hstack((lefthalf, lefthalf[:-1][::-1]))
            else:
                righthalf = (self.mat[i, w:-1] + self.mat[i, :w][::-1]) * 0.5
                self.mat[i, :] = np.hstack((righthalf[::-1], righthalf, self.mat[i, -1]))

    def norm_y(self, dist):
        """rescale every row so that the insert-size marginal equals `dist` (pyatac/VMat.py:101-104)"""
        for i in range(self.mat.shape[0]):
            self.mat[i] = self.mat[i] * (dist.get(size=i + self.lower) / np.sum(self.mat[i]))

    def smooth(self, sd=1):
        """2-D Gaussian smoothing with zero padding (pyatac/VMat.py:87-90: ndimage gaussian_filter, mode='constant')"""
        from scipy import ndimage
        self.mat = ndimage.gaussian_filter(self.mat, sd, mode="constant")

    def norm(self):
        """signal minus even background = 1 / (bases in window) x 10 (pyatac/VMat.py:95-100)"""
        tmp1 = self.mat / np.sum(self.mat)
        tmp2 = np.ones(self.mat.shape) * (1.0 / self.mat.size)
        self.mat = self.mat / (np.sum(self.mat * tmp1) - np.sum(self.mat * tmp2))
        self.mat = (self.mat / self.mat.shape[1]) * 10.0

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
    def standard():
        """the reference's raw standard V-plot (S. cer., insert sizes [0, 300), 501 columns): `vprocess --vplot` default"""
        return VMat.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "standard_vplot.npz"))

    @staticmethod
    def default():
        """the processed default V-plot (vprocess defaults: lower 105, upper 251, flank 60, smooth 0.75)"""
        return VMat.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data", "default_vmat.npz"))
