"""Nucleosome occupancy (API of the reference's nucleoatac/Occupancy.py:21-253).

Numerics: the windowed grid MLE, its smoothing and the coverage run in libnatac_hip.so (natac_run_occ);
`calculateOccupancy` on explicit vectors is natac_calculate_occupancy.  `modelNFR` (one global fit per run,
Occupancy.py:29-66) stays on the host with scipy, as do peak calling and `getNucDist`.
"""
import numpy as np
from scipy import optimize, stats
from scipy.special import gamma

from .. import get_context
from ..pipeline import BatchRunner, pack, window_size_hist
from ..pyatac.bias import PWM
from ..pyatac.chunk import Chunk
from ..pyatac.fragmentsizes import FragmentSizes
from ..pyatac.tracks import CoverageTrack, Track
from ..pyatac.utils import call_peaks, read_chrom_sizes_from_fasta, smooth


class FragmentMixDistribution(object):
    """insert-size distribution split into nucleosome-free and nucleosomal parts"""

    def __init__(self, lower=0, upper=2000):
        self.lower = lower
        self.upper = upper

    def getFragmentSizes(self, bamfile, chunklist=None):
        self.fragmentsizes = FragmentSizes(self.lower, self.upper)
        self.fragmentsizes.calculateSizes(bamfile, chunks=chunklist)

    def modelNFR(self, boundaries=(35, 115)):
        """gamma fit of the sub-nucleosomal sizes, extrapolated under the nucleosomal peak (Occupancy.py:29-66)"""
        fs = self.fragmentsizes
        head = fs.get(self.lower, boundaries[1])
        peak = int(np.where(head == max(head))[0][0]) + self.lower
        lo, hi = min(boundaries[0], peak), boundaries[1]
        x = np.arange(lo, hi)
        y = fs.get(lo, hi)

        def gamma_fit(X, o, p):
            k, theta, a = p[0], p[1], p[2]
            xm = X - o
            res = np.zeros(len(xm))
            nz = xm >= 0 if k >= 1 else xm > 0
            res[nz] = a * xm[nz] ** (k - 1) * np.exp(-xm[nz] / theta) / (theta ** k * gamma(k))
            return res

        score = np.ones(lo + 1) * float("inf")
        param = [0] * (lo + 1)
        pranges = ((0.01, 10), (0.01, 150), (0.01, 1))
        # scipy.optimize.brute(f, pranges, finish=fmin) of the reference (Occupancy.py:48-53), with its 20 x 20 x 20 grid evaluated
        # in one vectorised pass per offset instead of 8,000 Python calls (168,000 in total: ~2 s of a run's fixed cost): the same
        # grid (np.mgrid with complex steps, as brute builds it), the same element-wise formula, the first minimum, then brute's
        # own finisher from that grid point.
        grid = np.mgrid[tuple(slice(a, b, complex(20)) for a, b in pranges)]
        K, TH, A = (g.ravel()[:, None] for g in grid)
        norm_c = TH ** K * gamma(K)
        for i in range(15, lo + 1):
            f = lambda p: np.sum((gamma_fit(x, i, p) - y) ** 2)      # noqa: E731
            xm = (x - i)[None, :].astype(np.float64)
            nz = np.where(K >= 1, xm >= 0, xm > 0)
            with np.errstate(all="ignore"):
                vals = A * np.where(nz, xm, 1.0) ** (K - 1) * np.exp(-xm / TH) / norm_c
            J = np.sum((np.where(nz, vals, 0.0) - y[None, :]) ** 2, axis=1)
            j0 = int(np.argmin(J))
            x0 = np.array([K[j0, 0], TH[j0, 0], A[j0, 0]])
            res = optimize.fmin(f, x0, full_output=1, disp=0)
            score[i], param[i] = res[1], res[0]
        best = int(np.argmin(score))
        self.nfr_fit0 = FragmentSizes(self.lower, self.upper, vals=gamma_fit(np.arange(self.lower, self.upper), best, param[best]))
        nfr = np.concatenate((fs.get(self.lower, hi), self.nfr_fit0.get(hi, self.upper)))
        nfr[nfr == 0] = min(nfr[nfr != 0]) * 0.01
        self.nfr_fit = FragmentSizes(self.lower, self.upper, vals=nfr)
        nuc = np.concatenate((np.zeros(hi - self.lower), fs.get(hi, self.upper) - self.nfr_fit.get(hi, self.upper)))
        nuc[nuc <= 0] = min(min(nfr) * 0.1, min(nuc[nuc > 0]) * 0.001)
        self.nuc_fit = FragmentSizes(self.lower, self.upper, vals=nuc)


class OccupancyCalcParams(object):
    """alpha grid + normalised nuc / nfr size distributions + chi2 cutoff (Occupancy.py:89-102)"""

    def __init__(self, lower, upper, insert_dist, ci=0.9):
        self.lower = lower
        self.upper = upper
        nuc = insert_dist.nuc_fit.get(lower, upper)
        self.nuc_probs = nuc / np.sum(nuc)
        nfr = insert_dist.nfr_fit.get(lower, upper)
        self.nfr_probs = nfr / np.sum(nfr)
        self.alphas = np.linspace(0, 1, 101)
        self.l = len(self.alphas)
        self.cutoff = stats.chi2.ppf(ci, 1)

    def install(self, ctx, step=5, flank=60):
        ctx.set_occ_model(self.nuc_probs, self.nfr_probs, self.alphas, self.cutoff, step=step, flank=flank)


def calculateOccupancy(inserts, bias, params):
    """(occ, lower, upper) for one window of insert-size counts (Occupancy.py:104-120) -- GPU, literal formula"""
    ctx = get_context()
    params.install(ctx)
    return ctx.calculate_occupancy(np.asarray(inserts, dtype=np.float64), np.asarray(bias, dtype=np.float64))


class OccupancyTrack(Track):
    def __init__(self, chrom, start, end):
        Track.__init__(self, chrom, start, end, "occupancy")

    def calculateOccupancyMLE(self, mat, bias_mat, params):
        """occupancy on explicit matrices, one window every `step` bases (Occupancy.py:128-146)"""
        offset = self.start - mat.start
        if offset < params.flank:
            raise Exception("For calculateOccupancyMLE, mat does not have sufficient flanking regions")
        n = self.end - self.start
        self.vals = np.ones(n) * float("nan")
        self.lower_bound = np.ones(n) * float("nan")
        self.upper_bound = np.ones(n) * float("nan")
        ctx = get_context()
        params.occ_calc_params.install(ctx, step=params.step, flank=params.flank)
        for i in range(params.halfstep, n, params.step):
            a, b = self.start + i - params.flank, self.start + i + params.flank + 1
            ins = np.sum(mat.get(lower=0, upper=params.upper, start=a, end=b), axis=1)
            if np.sum(ins) > 0:
                bias = np.sum(bias_mat.get(lower=0, upper=params.upper, start=a, end=b), axis=1)
                left, right = i - params.halfstep, min(i + params.halfstep + 1, n)
                self.vals[left:right], self.lower_bound[left:right], self.upper_bound[left:right] = \
                    ctx.calculate_occupancy(ins, bias)

    def makeSmoothed(self, window_len=121, sd=20):
        kw = dict(window="gaussian", sd=sd, mode="same", norm=True)
        self.smoothed_vals = smooth(self.vals, window_len, **kw)
        self.smoothed_lower = smooth(self.lower_bound, window_len, **kw)
        self.smoothed_upper = smooth(self.upper_bound, window_len, **kw)


class OccPeak(Chunk):
    """one occupancy peak (Occupancy.py:155-171)"""

    def __init__(self, pos, chunk):
        self.chrom = chunk.chrom
        self.start = pos
        self.end = pos + 1
        self.strand = "*"
        i = pos - chunk.occ.start
        self.occ = chunk.occ.smoothed_vals[i]
        self.occ_lower = chunk.occ.smoothed_lower[i]
        self.occ_upper = chunk.occ.smoothed_upper[i]
        self.reads = chunk.cov.get(pos=pos)

    @staticmethod
    def from_values(chrom, pos, occ, occ_lower, occ_upper, reads):
        """an OccPeak whose values were gathered on the device (natac_run_occ_peaks)"""
        p = OccPeak.__new__(OccPeak)
        p.chrom, p.start, p.end, p.strand = chrom, pos, pos + 1, "*"
        p.occ, p.occ_lower, p.occ_upper, p.reads = float(occ), float(occ_lower), float(occ_upper), float(reads)
        return p

    def asBed(self):
        from ..pyatac.tracks import _py2_float_str as s
        return "\t".join([str(self.chrom), str(self.start), str(self.end), s(self.occ), s(self.occ_lower),
                          s(self.occ_upper), s(self.reads)])

    def write(self, handle):
        handle.write(self.asBed() + "\n")


class OccupancyParameters(object):
    """run-level occupancy parameters (Occupancy.py:175-193); like the reference, `fasta` is required here"""

    def __init__(self, insert_dist, upper, fasta, pwm, sep=120, min_occ=0.1, flank=60, out=None, bam=None, ci=0.9,
                 step=5):
        self.sep = sep
        self.chrs = read_chrom_sizes_from_fasta(fasta)
        self.fasta = fasta
        if fasta is not None:
            self.pwm = PWM.open(pwm)
        self.window = flank * 2 + 1
        self.min_occ = min_occ
        self.flank = flank
        self.bam = bam
        self.upper = upper
        self.occ_calc_params = OccupancyCalcParams(0, upper, insert_dist, ci=ci)
        if step % 2 == 0:
            step -= 1
        self.step = step
        self.halfstep = (self.step - 1) // 2


def occ_batch(chunks, params, ctx=None, with_flat=False):
    """OccChunk.process for a whole list of chunks in one GPU batch; returns the processed OccChunk objects
    (with_flat: also the concatenated per-base arrays + offsets, for the native track writer)"""
    ctx = ctx or get_context()
    params.occ_calc_params.install(ctx, step=params.step, flank=params.flank)
    pk = pack(chunks, params.bam, params.fasta, params.chrs, params.pwm if params.fasta is not None else None,
              window=params.window, upper=params.upper)
    run = BatchRunner(pk, ctx)
    try:
        res = run.occ()
        # OccChunk.callPeaks + getNucDist (Occupancy.py:225-240) for every chunk on the device: peak search, OccPeak values,
        # the occ_lower > min_occ / reads > 0 filter and the per-chunk nucleosomal size distribution
        pk_chunk, pk_pos, p_occ, p_lo, p_up, p_rd, p_keep, nuc_dist = run.batch.run_occ_peaks(min_occ=params.min_occ, sep=params.sep)
        # chunks with more local maxima than the device peak finder holds per chunk: utils.call_peaks on the host
        host_peaks = {}
        for k in np.nonzero(run.batch.status() & 2)[0]:
            host_peaks[int(k)] = call_peaks(res["smoothed_vals"][k].copy(), min_signal=params.min_occ, sep=params.sep,
                                            boundary=params.sep // 2, order=1)
    finally:
        run.close()
    bounds = np.searchsorted(pk_chunk, np.arange(len(chunks) + 1))
    out = []
    for k, ch in enumerate(chunks):
        oc = OccChunk(ch)
        oc.params = params
        oc._pk, oc._k = pk, k
        oc.occ = OccupancyTrack(ch.chrom, ch.start, ch.end)
        oc.occ.vals, oc.occ.lower_bound, oc.occ.upper_bound = res["vals"][k], res["lower_bound"][k], res["upper_bound"][k]
        # smoothed_vals as natac_run_occ leaves it == after call_peaks' in-place NaN fill (utils.py:86-91)
        oc.occ.smoothed_vals = res["smoothed_vals"][k].copy()
        oc.occ.smoothed_lower = res["smoothed_lower"][k].copy()
        oc.occ.smoothed_upper = res["smoothed_upper"][k].copy()
        oc.cov = CoverageTrack(ch.chrom, ch.start, ch.end)
        oc.cov.vals = res["cov"][k].copy()
        if k in host_peaks:
            oc.callPeaks(peaks=host_peaks[k])          # nuc_dist of these chunks: host histogram in getNucDist
        else:
            a, e = int(bounds[k]), int(bounds[k + 1])
            for i in range(a, e):
                if p_keep[i]:
                    oc.peaks[int(pk_pos[i])] = OccPeak.from_values(ch.chrom, int(pk_pos[i]) + ch.start, p_occ[i], p_lo[i], p_up[i], p_rd[i])
            oc._nuc_dist = nuc_dist[k].copy()
        out.append(oc)
    if with_flat:
        # smoothed_vals was NaN-filled on the device exactly like call_peaks does in place, so the flat array is what
        # the reference's writer sees (run_occ.py:47)
        return out, dict(out_off=pk.out_off, **run.flat)
    return out


class OccChunk(Chunk):
    """occupancy of one chunk (Occupancy.py:195-253)"""

    def __init__(self, chunk):
        self.start = chunk.start
        self.end = chunk.end
        self.chrom = chunk.chrom
        self.peaks = {}
        self.nfrs = []

    def callPeaks(self, peaks=None):
        """occupancy peaks (Occupancy.py:225-231); `peaks` = positions already found on the device by occ_batch"""
        if peaks is None:
            peaks = call_peaks(self.occ.smoothed_vals, sep=self.params.sep, min_signal=self.params.min_occ)
        for peak in peaks:
            tmp = OccPeak(int(peak) + self.start, self)
            if tmp.occ_lower > self.params.min_occ and tmp.reads > 0:
                self.peaks[int(peak)] = tmp

    def getNucDist(self):
        """insert-size distribution around the called peaks (Occupancy.py:232-240): computed on the device with the peaks
        (natac_run_occ_peaks); the host histogram remains for chunks whose peaks were found on the host"""
        if getattr(self, "_nuc_dist", None) is not None:
            return self._nuc_dist
        nuc_dist = np.zeros(self.params.upper)
        for peak in self.peaks.keys():
            h = window_size_hist(self._pk, self._k, self.peaks[peak].start - self.start, self.params.flank, self.params.upper)
            nuc_dist += h / float(np.sum(h))
        return nuc_dist

    def process(self, params):
        """fragment matrix -> bias -> occupancy -> coverage -> peaks, on the GPU (a batch of one chunk)"""
        done = occ_batch([Chunk(self.chrom, self.start, self.end)], params)[0]
        self.__dict__.update(done.__dict__)

    def removeData(self):
        for name in list(self.__dict__.keys()):
            delattr(self, name)
