"""Tn5 insertion bias (API of the reference's pyatac/bias.py:30-124): PWM descriptor + per-base log-bias track."""
import os

import numpy as np

from . import seq
from .tracks import Track

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")


BUILTIN_PWMS = ("Human", "Human2", "Yeast")   # Tn5 preference tables shipped in data/tn5_pwm_tables.npz


def pwm_parse(name):
    """built-in table name or path of a PWM descriptor file (pyatac/bias.py:20-27)"""
    return name


class PWM(object):
    def __init__(self, mat, up, down, nucleotides):
        self.mat = mat
        self.up = up
        self.down = down
        self.nucleotides = nucleotides

    def save(self, filename):
        with open(filename, "w") as out:
            out.write("#PWM Descriptor File\n#Contains PWM and pertinent information\n")
            out.write("#up\n%d\n#down\n%d\n#nucleotides\n%s\n#mat\n" % (self.up, self.down, "\t".join(self.nucleotides)))
            for row in self.mat:
                out.write("\t".join(repr(float(x)) for x in row) + "\n")

    @staticmethod
    def open(name):
        if name in BUILTIN_PWMS:
            t = np.load(os.path.join(_DATA, "tn5_pwm_tables.npz"), allow_pickle=False)
            return PWM(np.array(t[name + "_mat"]), int(t[name + "_up"]), int(t[name + "_down"]),
                       [str(x) for x in t[name + "_nucleotides"]])
        state, d, rows = "", {}, []
        with open(pwm_parse(name)) as f:
            for line in f:
                if line.startswith("#"):
                    key = line.strip().lstrip("#")
                    state = key if key in ("up", "down", "mat", "nucleotides") else "other"
                elif state in ("up", "down"):
                    d[state] = int(line.strip())
                elif state == "nucleotides":
                    d["nucleotides"] = line.strip("\n").split()
                elif state == "mat":
                    rows.append([float(x) for x in line.strip("\n").split("\t") if x != ""])
        if not all(k in d for k in ("up", "down", "nucleotides")) or not rows:
            raise Exception("PWM decriptor file appeas to be missing some needed components")
        return PWM(np.array(rows), d["up"], d["down"], d["nucleotides"])


class InsertionBiasTrack(Track):
    def __init__(self, chrom, start, end, log=True):
        Track.__init__(self, chrom, start, end, name="insertion bias", log=log)

    def computeBias(self, fasta, chromDict, pwm):
        """log PWM score of every position from the genome sequence (pyatac/bias.py:85-92) -- natac_pwm_bias"""
        from .. import get_context
        self.slop(chromDict, up=pwm.up, down=pwm.down)
        sequence = seq.get_sequence(self, fasta)
        self.vals = get_context().pwm_bias(sequence, pwm.mat, pwm.nucleotides)
        self.start += pwm.up
        self.end -= pwm.down

    def get(self, start=None, end=None, pos=None, log=None):
        out = Track.get(self, start, end, pos)
        if log is None or bool(log) == bool(self.log):
            return out
        return np.log(out) if log else np.exp(out)
