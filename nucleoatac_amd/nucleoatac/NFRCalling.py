"""Nucleosome-free regions between neighbouring nucleosome calls (API of the reference's nucleoatac/NFRCalling.py).

The per-base inputs come from the accelerated path: the insertion track of every chunk is natac_run_ins on one batch, the
bias track is the PWM kernel (InsertionBiasTrack.computeBias), the occupancy tracks are read back through the native
tabix reader.  The interval logic itself (a handful of means per chunk) is host code like the reference's."""
import numpy as np

from ..pyatac.bias import PWM, InsertionBiasTrack
from ..pyatac.chunk import Chunk
from ..pyatac.tracks import InsertionTrack, Track, _py2_float_str
from ..pyatac.utils import read_chrom_sizes_from_fasta
from ..tabix import TabixFile


class NFR(Chunk):
    """one NFR (NFRCalling.py:16-32): mean occupancy, min of the upper bound, insertion and bias densities"""

    def __init__(self, left, right, nfrtrack):
        self.chrom = nfrtrack.chrom
        self.start = left
        self.end = right
        self.strand = "*"
        self.occ = np.mean(nfrtrack.occ.get(left, right))
        self.min_upper = np.min(nfrtrack.occ_upper.get(left, right))
        self.ins_density = np.mean(nfrtrack.ins.get(left, right))
        # the reference needs --fasta here (its bias track has no values otherwise and NFR() raises); without one the
        # column is written as nan
        self.bias_density = np.mean(nfrtrack.bias.get(left, right, log=False)) if nfrtrack.bias.vals is not None else np.nan

    def asBed(self):
        return "\t".join([str(self.chrom), str(self.start), str(self.end)] +
                         [_py2_float_str(float(x)) for x in (self.occ, self.min_upper, self.ins_density, self.bias_density)])

    def write(self, handle):
        handle.write(self.asBed() + "\n")


class NFRParameters(object):
    """NFRCalling.py:35-48"""

    def __init__(self, occ_track, calls, ins_track=None, bam=None, max_occ=0.25, max_occ_upper=0.25, fasta=None, pwm=None):
        self.bam = bam
        self.ins_track = ins_track
        self.occ_track = occ_track
        self.calls = calls
        self.max_occ = max_occ
        self.max_occ_upper = max_occ_upper
        self.fasta = fasta
        if fasta is not None:
            self.pwm = PWM.open(pwm)
            self.chrs = read_chrom_sizes_from_fasta(fasta)


class NFRChunk(Chunk):
    """NFR calls of one chunk (NFRCalling.py:52-118)"""

    def __init__(self, chunk):
        self.start = chunk.start
        self.end = chunk.end
        self.chrom = chunk.chrom
        self.nfrs = []

    def initialize(self, parameters):
        self.params = parameters

    def getOcc(self):
        self.occ = Track(self.chrom, self.start, self.end, "Occupancy")
        self.occ.read_track(self.params.occ_track)
        upper_file = self.params.occ_track[:-11] + "upper_bound.bedgraph.gz"
        self.occ_upper = Track(self.chrom, self.start, self.end, "Occupancy")
        self.occ_upper.read_track(upper_file)

    def getIns(self, vals=None):
        """insertion track: from the GPU batch (`vals`), computed for this chunk alone, or read from --ins_track"""
        if self.params.ins_track is None:
            self.ins = InsertionTrack(self.chrom, self.start, self.end)
            if vals is not None:
                self.ins.vals = vals
            else:
                self.ins.calculateInsertions(self.params.bam)
        else:
            self.ins = Track(self.chrom, self.start, self.end, "Insertion")
            self.ins.read_track(self.params.ins_track)

    def getBias(self):
        self.bias = InsertionBiasTrack(self.chrom, self.start, self.end, log=True)
        if self.params.fasta is not None:
            self.bias.computeBias(self.params.fasta, self.params.chrs, self.params.pwm)

    def findNFRs(self):
        """regions between consecutive calls, 73 / 72 bp off the dyads, that are depleted of nucleosomes (:96-110)"""
        tbx = TabixFile(self.params.calls)
        nucs = []
        if self.chrom in tbx.contigs:
            for row in tbx.fetch(self.chrom, self.start, self.end):
                nucs.append(int(row.split("\t")[1]))
        tbx.close()
        for j in range(1, len(nucs)):
            left = nucs[j - 1] + 73
            right = nucs[j] - 72
            if right <= left:
                continue
            candidate = NFR(left, right, self)
            if candidate.min_upper < self.params.max_occ_upper and candidate.occ < self.params.max_occ:
                self.nfrs.append(candidate)

    def process(self, params, ins_vals=None):
        self.initialize(params)
        self.getOcc()
        self.getIns(ins_vals)
        self.getBias()
        self.findNFRs()

    def removeData(self):
        for name in list(self.__dict__.keys()):
            delattr(self, name)
