"""Nucleosome signal + calling (API of the reference's nucleoatac/NucleosomeCalling.py:25-345).

GPU (libnatac_hip.so): coverage, raw signal, background, normalised + smoothed signal (natac_run_nuc), per-candidate
log-likelihood ratio / multinomial variance / z (natac_run_candidates), dense correlate for the operator-level
SignalTrack / BiasTrack classes.  Host: peak calling, thresholds, the L-BFGS fuzziness fit.
"""
import os
from bisect import bisect_left
from copy import copy

import numpy as np
from scipy import optimize

from .. import get_context
from ..pipeline import BatchRunner, pack
from ..pyatac.bias import PWM
from ..pyatac.chunk import Chunk
from ..pyatac.tracks import CoverageTrack, Track, _py2_float_str
from ..pyatac.utils import call_peaks, read_chrom_sizes_from_bam, reduce_peaks


class SignalTrack(Track):
    """V-plot cross-correlation signal (NucleosomeCalling.py:25-36)"""

    def __init__(self, chrom, start, end):
        Track.__init__(self, chrom, start, end, "signal")

    def calculateSignal(self, mat, vmat):
        offset = self.start - mat.start - vmat.w
        if offset < 0:
            raise Exception("Insufficient flanking region on mat to calculate signal")
        sub = mat.get(vmat.lower, vmat.upper, mat.start + offset, mat.end - offset)
        self.vals = get_context().correlate_valid(sub, vmat.mat)


class NormSignalTrack(Track):
    def __init__(self, chrom, start, end):
        Track.__init__(self, chrom, start, end, "normalized signal")

    def calculateNormSignal(self, raw, bias):
        self.vals = raw.get(self.start, self.end) - bias.get(self.start, self.end)


class BiasTrack(Track):
    """background model of the signal (NucleosomeCalling.py:45-64)"""

    def __init__(self, chrom, start, end):
        Track.__init__(self, chrom, start, end, "bias")

    def calculateBackgroundSignal(self, mat, vmat, nuc_cov):
        offset = self.start - mat.start - vmat.w
        if offset < 0:
            raise Exception("Insufficient flanking region on mat to calculate signal")
        self.vmat = vmat
        self.bias_mat = mat
        self.cov = CoverageTrack(self.chrom, self.start, self.end)
        self.cov.calculateCoverage(self.bias_mat, vmat.lower, vmat.upper, vmat.w * 2 + 1)
        self.nuc_cov = nuc_cov.vals
        sub = self.bias_mat.get(vmat.lower, vmat.upper, self.bias_mat.start + offset, self.bias_mat.end - offset)
        with np.errstate(invalid="ignore", divide="ignore"):
            self.vals = get_context().correlate_valid(sub, vmat.mat) * self.nuc_cov / self.cov.vals


class SignalDistribution(object):
    """distribution of the signal at one position under the background model (NucleosomeCalling.py:68-88)"""

    def __init__(self, position, vmat, bias_mat, reads):
        self.position = position
        self.reads = reads
        self.vmat = vmat
        sub = bias_mat.get(vmat.lower, vmat.upper, position - vmat.w, position + vmat.w + 1)
        self.prob_mat = sub / np.sum(sub)
        self.probs = self.prob_mat.flatten()

    def simulateReads(self):
        return np.reshape(np.random.multinomial(self.reads, self.probs), self.vmat.mat.shape)

    def simulateDist(self, numiters=1000):
        self.scores = [np.sum(self.simulateReads() * self.vmat.mat) for _ in range(numiters)]

    def analStd(self):
        """sqrt of calculateCov (nucleoatac/multinomial_cov.pyx:20-31) -- natac_calculate_cov"""
        return np.sqrt(get_context().calculate_cov(self.probs, np.ravel(self.vmat.mat), int(self.reads)))

    def analMean(self):
        return np.sum(self.prob_mat * self.vmat.mat * self.reads)


def norm(x, v, w, mean):
    """normal pdf with variance v scaled to height w (NucleosomeCalling.py:92-97)"""
    y = 1.0 / np.sqrt(2 * np.pi * v) * np.exp(-(x - mean) ** 2 / (2 * v))
    return y * (w / y.max())


_OCC_POOL = None


def occ_reader_pool():
    """ONE module-level thread pool for the tabix region reads of NucChunk.getOcc (NucleosomeCalling.py:284-293): the readers
    are thread-local (pyatac/tracks.py:_tabix), so keeping the threads alive across batches keeps every thread's three parsed
    .tbi indexes alive too, instead of re-opening and re-parsing them in up to 16 new threads per batch."""
    global _OCC_POOL
    if _OCC_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _OCC_POOL = ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1)), thread_name_prefix="natac-occ-reader")
    return _OCC_POOL


def map_in_slices(fn, items, slices_per_thread=2):
    """[fn(x) for x in items] on the reader pool, every thread taking CONTIGUOUS runs of `items`: the native tabix reader of a
    thread keeps the members it inflated last (natac_tabix.hpp), and neighbouring regions share most of them"""
    items = list(items)
    pool = occ_reader_pool()
    n_slices = max(1, min(len(items), pool._max_workers * slices_per_thread))
    if n_slices < 2:
        return [fn(x) for x in items]
    cuts = [len(items) * i // n_slices for i in range(n_slices + 1)]
    parts = pool.map(lambda ab: [fn(x) for x in items[ab[0]:ab[1]]], zip(cuts[:-1], cuts[1:]))
    return [r for part in parts for r in part]


def read_occ_tracks(occ_track, chrom, start, end):
    """(occ, lower, upper) values of [start, end) from the three track files `occ` wrote (NucChunk.getOcc)"""
    base = occ_track[:-11]
    out = []
    for f in (occ_track, base + "lower_bound.bedgraph.gz", base + "upper_bound.bedgraph.gz"):
        t = Track(chrom, start, end, "Occupancy")
        t.read_track(f)
        out.append(t.vals)
    return out


def read_regions_of(path, chunks, value_col=4):
    """Track.read_track of every chunk from ONE indexed file in one native call (natac_tbx_read_regions: the chunks of a batch
    are position-sorted, its cursors inflate and parse every BGZF member once): (flat values, offsets), or None when the file
    has no index / the native reader is not there (the callers then read chunk by chunk)"""
    from ..pyatac.tracks import _tabix
    if not (path.endswith(".gz") and os.path.exists(path + ".tbi")):
        return None
    rd = _tabix(path)
    if not hasattr(rd, "read_regions"):
        return None
    return rd.read_regions([c.chrom for c in chunks], [c.start for c in chunks], [c.end for c in chunks], value_col=value_col)


def read_occ_tracks_many(occ_track, chunks):
    """read_occ_tracks for a list of chunks: per chunk [occ, lower, upper].  A missing / damaged track file or index RAISES, as
    NucChunk.getOcc does in the reference (NucleosomeCalling.py:284-293: Track.read_track outside any try -- the run aborts);
    only the per-position lookup of Nucleosome.getOcc (:128-135) turns a failure into NaN, and that stays with the callers.
    Deviation, on purpose: the drivers call this for chunks WITH calls only (`batch_calls_start`), the reference reads the track of every
    chunk -- so the three files and their indexes are checked once up front here, for any chunk list, and a damaged member under a
    chunk without calls goes unnoticed (its values are never used)."""
    from .. import occstore
    res = occstore.lookup(occ_track)
    if res is not None and chunks:          # written by this process: the values are still in HBM, as the files show them
        from .. import get_context
        ctx, chroms = get_context(), [c.chrom for c in chunks]
        starts, ends = [c.start for c in chunks], [c.end for c in chunks]
        three = [res.read_regions(ctx, chroms, starts, ends, slot) for slot in range(3)]
        if all(t is not None for t in three):
            return [[flat[int(off[k]):int(off[k + 1])] for flat, off in three] for k in range(len(chunks))]
    base = occ_track[:-11]
    files = (occ_track, base + "lower_bound.bedgraph.gz", base + "upper_bound.bedgraph.gz")
    for f in files:
        if not os.path.exists(f):
            raise IOError("occupancy track %s not found (expected next to --occ_track %s)" % (f, occ_track))
    three = [read_regions_of(f, chunks) for f in files]
    if any(t is None for t in three):      # no index / no native reader: chunk by chunk through Track.read_track
        return map_in_slices(lambda ch: read_occ_tracks(occ_track, ch.chrom, ch.start, ch.end), chunks)
    return [[flat[int(off[k]):int(off[k + 1])] for flat, off in three] for k in range(len(chunks))]


FAST_FD = True    # batched finite differences in fit_fuzz_one (False: scipy's own numerical gradient)


def fit_fuzz_one(vals, allnucs, index, nonredundant_sep, smooth_sd):
    """Nucleosome.getFuzz of the reference (NucleosomeCalling.py:137-194) for the call at chunk-relative position `index`:
    up to three Gaussians (the call and its neighbours closer than nonredundant_sep) fitted to the smoothed signal with
    scipy's L-BFGS-B; returns (fuzz, weight, fit_pos).  Clamps vals[left:right] at 0 in place like the reference."""
    third = nonredundant_sep // 3
    x = bisect_left(allnucs, index)
    if x > 0 and index - allnucs[x - 1] < nonredundant_sep:
        left = allnucs[x - 1]
        means = (index - allnucs[x - 1], 0)
    else:
        left = index - third
        means = (third,)
    if x < len(allnucs) - 1 and allnucs[x + 1] - index < nonredundant_sep:
        right = allnucs[x + 1]
        means += (allnucs[x + 1] - left,)
    else:
        right = index + third + 1
    sig = vals[left:right]
    sig[sig < 0] = 0
    top = max(sig)
    bounds, guess = (), ()
    for m in means:
        bounds += ((2 ** 2, 50 ** 2), (0.001, top * 1.1), (m - 10, m + 10))
        guess += (smooth_sd ** 2, top * 0.9, m)
    xs = np.linspace(0, len(sig) - 1, len(sig))

    def err(pars):
        fit = np.zeros(len(xs))
        for j in range(len(pars) // 3):
            fit += norm(xs, pars[3 * j], pars[3 * j + 1], pars[3 * j + 2])
        return np.sum((fit - sig) ** 2)

    if not FAST_FD:
        res = optimize.minimize(err, guess, bounds=bounds, method="L-BFGS-B")
        return np.sqrt(res["x"][0]), res["x"][1], res["x"][2] + left

    # The reference lets scipy difference `err` numerically: n + 1 Python calls per gradient.  `fun_and_grad` forms the SAME 2-point
    # differences (absolute step 1e-8, flipped at an upper bound: scipy.optimize._numdiff.approx_derivative as L-BFGS-B
    # calls it) from one batched evaluation of the n shifted parameter vectors; every element goes through the same numpy
    # operations as in `err`, so the optimiser sees bit-identical values and takes the same path (tests/test_host_logic.py).
    lb = np.array([b[0] for b in bounds], dtype=np.float64)
    ub = np.array([b[1] for b in bounds], dtype=np.float64)
    n = len(guess)
    idx = np.arange(n)

    def fun_and_grad(pars):
        """(err(pars), its 2-point finite differences): one evaluation of the objective at pars shared by both -- L-BFGS-B always
        asks for the pair"""
        x0 = np.asarray(pars, dtype=np.float64)
        h = np.full(n, 1e-8)
        lower_dist, upper_dist = x0 - lb, ub - x0
        x = x0 + h
        violated = (x < lb) | (x > ub)
        fitting = np.abs(h) <= np.maximum(lower_dist, upper_dist)
        h[violated & fitting] *= -1
        forward = (upper_dist >= lower_dist) & ~fitting
        h[forward] = upper_dist[forward]
        backward = (upper_dist < lower_dist) & ~fitting
        h[backward] = -lower_dist[backward]
        X = np.tile(x0, (n, 1))
        X[idx, idx] = x0 + h
        dx = X[idx, idx] - x0
        fit = np.zeros((n, len(xs)))
        for j in range(n // 3):
            v, w, mean = X[:, 3 * j][:, None], X[:, 3 * j + 1][:, None], X[:, 3 * j + 2][:, None]
            y = 1.0 / np.sqrt(2 * np.pi * v) * np.exp(-(xs[None, :] - mean) ** 2 / (2 * v))
            fit += y * (w / y.max(axis=1)[:, None])
        r = (fit - sig[None, :]) ** 2
        f1 = np.array([np.sum(r[i]) for i in range(n)])
        f0 = err(x0)
        return f0, (f1 - f0) / dx

    res = optimize.minimize(fun_and_grad, guess, jac=True, bounds=bounds, method="L-BFGS-B")
    return np.sqrt(res["x"][0]), res["x"][1], res["x"][2] + left


LOCKSTEP = True   # advance all fits of a task together (fuzzfit.py) when this scipy allows it; False: one call at a time


def fit_fuzz_chunk(task):
    """all calls of one chunk in ascending order; task = (smoothed values, sorted call positions, nonredundant_sep, smooth_sd)"""
    return fit_fuzz_chunks([task])[0]


def fit_fuzz_chunks(tasks):
    """the calls of several chunks (the unit of work of the --cores pool): one list of (fuzz, weight, fit_pos) per task.
    The fits of all tasks advance in lockstep (fuzzfit.fit_many: bit-identical to the one-at-a-time path, ~4x faster)."""
    from . import fuzzfit
    lock = LOCKSTEP and FAST_FD and fuzzfit.available()
    out, probs = [], []
    for vals, keys, nonredundant_sep, smooth_sd in tasks:
        vals = np.array(vals, dtype=np.float64)
        keys = [int(k) for k in keys]
        if lock:
            probs += [fuzzfit.problem(vals, keys, k, nonredundant_sep, smooth_sd) for k in keys]
            out.append(len(keys))
        else:
            out.append([fit_fuzz_one(vals, keys, k, nonredundant_sep, smooth_sd) for k in keys])
    if lock:
        res = fuzzfit.fit_many(probs)
        ends = np.cumsum(out)
        out = [[tuple(r) for r in res[e - c:e]] for c, e in zip(out, ends)]
    return out


def fit_fuzz_tasks(tasks, pool=None, pool_workers=1, start_only=False):
    """fit_fuzz_chunks over a list of per-chunk tasks, in slices over the process pool when there is one (4 slices per
    worker: the lockstep groups stay full, the tail of slow fits stays short); results in task order.  start_only: submit
    the slices and return a function that waits for the results (the caller reads its occupancy tracks meanwhile)."""
    if pool is None or len(tasks) < 2:
        res = fit_fuzz_chunks(tasks)
        return (lambda: res) if start_only else res
    per = max(1, -(-len(tasks) // (4 * pool_workers)))
    parts = pool.map(fit_fuzz_chunks, [tasks[a:a + per] for a in range(0, len(tasks), per)])   # submitted here
    collect = lambda: [r for part in parts for r in part]      # noqa: E731
    return collect if start_only else collect()


class Nucleosome(Chunk):
    """one candidate / called nucleosome (NucleosomeCalling.py:99-202)"""

    def __init__(self, pos, nuctrack):
        self.chrom = nuctrack.chrom
        self.start = pos
        self.end = pos + 1
        self.nfr_cov = nuctrack.nfr_cov.get(pos=pos)
        self.nuc_cov = nuctrack.nuc_cov.get(pos=pos)
        self.nuc_signal = nuctrack.nuc_signal.get(pos=pos)
        self.norm_signal = nuctrack.norm_signal.get(pos=pos)
        self.smoothed = nuctrack.smoothed.get(pos=pos)

    def getOcc(self, nuctrack):
        try:
            self.occ = nuctrack.occ.get(pos=self.start)
            self.occ_lower = nuctrack.occ_lower.get(pos=self.start)
            self.occ_upper = nuctrack.occ_upper.get(pos=self.start)
        except Exception:
            self.occ = self.occ_lower = self.occ_upper = np.nan

    def getFuzz(self, nuctrack):
        """sd of a (mixture of up to 3) Gaussian(s) fitted to the smoothed signal around the call
        (NucleosomeCalling.py:137-194; scipy L-BFGS-B on the host)"""
        p = nuctrack.params
        self.fuzz, self.weight, self.fit_pos = fit_fuzz_one(nuctrack.smoothed.vals, nuctrack.sorted_nuc_keys,
                                                            self.start - nuctrack.start, p.nonredundant_sep, p.smooth_sd)

    def asBed(self):
        s = _py2_float_str
        return "\t".join([str(self.chrom), str(self.start), str(self.end)] + [
            s(float(v)) for v in (self.z, self.occ, self.occ_lower, self.occ_upper, self.lr, self.norm_signal,
                                  self.nuc_signal, self.nuc_cov, self.nfr_cov, self.fuzz)])

    def write(self, handle):
        handle.write(self.asBed() + "\n")


class NucParameters(object):
    """run-level parameters of nucleosome calling (NucleosomeCalling.py:204-226)"""

    def __init__(self, vmat, fragmentsizes, bam, fasta, pwm, occ_track=None, atac=True, sd=25, nonredundant_sep=120,
                 redundant_sep=25, min_z=3, min_lr=0, min_reads=1):
        self.atac = atac
        self.vmat = vmat
        self.lower = vmat.lower
        self.upper = vmat.upper
        self.window = vmat.mat.shape[1]
        self.fragmentsizes = fragmentsizes
        self.min_reads = min_reads
        self.min_z = min_z
        self.min_lr = min_lr
        self.smooth_sd = sd
        self.redundant_sep = redundant_sep
        self.nonredundant_sep = nonredundant_sep
        self.fasta = fasta
        self.pwm = PWM.open(pwm)
        self.chrs = read_chrom_sizes_from_bam(bam)
        self.bam = bam
        self.occ_track = occ_track

    def install(self, ctx):
        ctx.set_vmat(self.vmat.mat, self.vmat.lower, self.vmat.upper)
        ctx.set_sizes(self.fragmentsizes.get(0, self.vmat.upper))


def nuc_batch(chunks, params, ctx=None, with_flat=False):
    """NucChunk.process for a list of chunks in one GPU batch; returns the processed NucChunk objects"""
    ctx = ctx or get_context()
    params.install(ctx)
    if params.fasta is not None:
        from ..pyatac.utils import read_chrom_sizes_from_fasta
        chrs = read_chrom_sizes_from_fasta(params.fasta)
    else:
        chrs = params.chrs
    pk = pack(chunks, params.bam, params.fasta, chrs, params.pwm, atac=params.atac, window=params.window, upper=params.upper)
    run = BatchRunner(pk, ctx)
    out = []
    try:
        res = run.nuc(params.smooth_sd)
        for k, ch in enumerate(chunks):
            nc = NucChunk(ch)
            nc.initialize(params)
            for name, cls in (("nuc_cov", CoverageTrack), ("nfr_cov", CoverageTrack)):
                t = cls(ch.chrom, ch.start, ch.end)
                t.vals = res[name][k].copy()
                setattr(nc, name, t)
            for name, label in (("nuc_signal", "signal"), ("bias", "bias"), ("norm_signal", "normalized signal"),
                                ("smoothed", "Smooth Signal")):
                setattr(nc, name, Track(ch.chrom, ch.start, ch.end, label, vals=res[name][k].copy()))
            out.append(nc)
        if params.occ_track is not None:
            # three tabix region reads per chunk (NucleosomeCalling.py:284-293), ~1 ms each: spread over host threads (the
            # native reader releases the GIL; every thread has its own readers, pyatac/tracks.py:_tabix)
            map_in_slices(NucChunk.getOcc, out)
        # candidate search (call_peaks on norm + smoothed, NucleosomeCalling.py:297-301) and LR / z for every candidate
        # of every chunk on the device: nothing round-trips between the signal kernels and the statistics
        cc, cp, lr, var, z = run.batch.run_peaks(min_signal=0, sep=params.redundant_sep,
                                                 boundary=params.nonredundant_sep // 2, order=params.redundant_sep // 2)
        # chunks with more local maxima than the device peak finder holds per chunk: utils.call_peaks on the host, the
        # statistics of those candidates still on the device (natac_run_candidates)
        host = {}
        over = np.nonzero(run.batch.status() & 2)[0]
        if len(over):
            hc, hp = [], []
            for k in over:
                combined = res["norm_signal"][k] + res["smoothed"][k]
                pos = np.asarray(call_peaks(combined, min_signal=0, sep=params.redundant_sep,
                                            boundary=params.nonredundant_sep // 2, order=params.redundant_sep // 2), np.int32)
                hc.append(np.full(len(pos), k, np.int32))
                hp.append(pos)
            hc, hp = np.concatenate(hc), np.concatenate(hp)
            hlr, _hvar, hz = run.batch.run_candidates(hc, hp)
            hb = np.searchsorted(hc, np.arange(len(out) + 1))
            for k in over:
                a, b = int(hb[k]), int(hb[k + 1])
                host[int(k)] = (hp[a:b], hlr[a:b], hz[a:b])
        bounds = np.searchsorted(cc, np.arange(len(out) + 1))
        for k, nc in enumerate(out):
            a, b = int(bounds[k]), int(bounds[k + 1])
            if k in host:
                nc._cands, klr, kz = host[k]
            else:
                nc._cands, klr, kz = cp[a:b], lr[a:b], z[a:b]
            nc.findAllNucs(stats=(klr, kz))
        # the L-BFGS fits are independent per call: advanced in lockstep, farmed out like the reference's --cores pool
        tasks = [(nc.smoothed.vals, nc.sorted_nuc_keys, params.nonredundant_sep, params.smooth_sd) for nc in out]
        for nc, r in zip(out, fit_fuzz_tasks(tasks, getattr(params, "pool", None), getattr(params, "pool_workers", 1))):
            nc.fit(results=r)
    finally:
        run.close()
    if with_flat:
        return out, dict(out_off=pk.out_off, **run.flat)
    return out


class NucChunk(Chunk):
    """nucleosome signal + calls of one chunk (NucleosomeCalling.py:230-345)"""

    def __init__(self, chunk):
        self.start = chunk.start
        self.end = chunk.end
        self.chrom = chunk.chrom

    def initialize(self, parameters):
        self.params = parameters

    def getOcc(self):
        """occupancy tracks written by `occ` (NucleosomeCalling.py:284-293)"""
        base = self.params.occ_track[:-11]
        for attr, f in (("occ", self.params.occ_track), ("occ_lower", base + "lower_bound.bedgraph.gz"),
                        ("occ_upper", base + "upper_bound.bedgraph.gz")):
            t = Track(self.chrom, self.start, self.end, "Occupancy")
            t.read_track(f)
            setattr(self, attr, t)

    def candidatePositions(self):
        """local maxima of norm + smoothed signal (NucleosomeCalling.py:297-301)"""
        combined = self.norm_signal.vals + self.smoothed.vals
        return call_peaks(combined, min_signal=0, sep=self.params.redundant_sep,
                          boundary=self.params.nonredundant_sep // 2, order=self.params.redundant_sep // 2)

    def findAllNucs(self, stats=None):
        """threshold the candidates in the reference's order: reads, LR, z (NucleosomeCalling.py:294-315)"""
        if stats is None:
            raise Exception("findAllNucs needs the candidate statistics of natac_run_candidates (use nuc_batch / process)")
        lr, z = stats
        self.nuc_collection = {}
        for j, i in enumerate(self._cands):
            nuc = Nucleosome(int(i) + self.start, self)
            if nuc.nuc_cov > self.params.min_reads:
                nuc.lr = lr[j]
                if nuc.lr > self.params.min_lr:
                    nuc.z = z[j]
                    if nuc.z >= self.params.min_z:
                        if hasattr(self, "occ"):
                            nuc.getOcc(self)
                        else:
                            nuc.occ = nuc.occ_lower = nuc.occ_upper = np.nan
                        self.nuc_collection[int(i)] = nuc
        self.sorted_nuc_keys = np.array(sorted(self.nuc_collection.keys()))
        self.nonredundant = reduce_peaks(self.sorted_nuc_keys, [self.nuc_collection[x].z for x in self.sorted_nuc_keys],
                                         self.params.nonredundant_sep)
        self.redundant = np.setdiff1d(self.sorted_nuc_keys, self.nonredundant)

    def fit(self, results=None):
        """fuzziness of every call + the fitted signal (NucleosomeCalling.py:316-324); `results`: fit_fuzz_chunk output
        computed elsewhere (the --cores pool)"""
        x = np.linspace(0, self.length() - 1, self.length())
        fit = np.zeros(self.length())
        for j, k in enumerate(self.sorted_nuc_keys):
            n = self.nuc_collection[int(k)]
            if results is None:
                n.getFuzz(self)
            else:
                n.fuzz, n.weight, n.fit_pos = results[j]
            fit += norm(x, n.fuzz ** 2, n.weight, n.fit_pos)
        self.fitted = Track(self.chrom, self.start, self.end, "Fitted Nucleosome Signal")
        self.fitted.assign_track(fit)

    def process(self, params):
        """signal tracks + calls on the GPU (a batch of one chunk); makeInsertionTrack is skipped -- its result is
        discarded by the reference's _nucHelper (run_nuc.py:30-32)"""
        done = nuc_batch([Chunk(self.chrom, self.start, self.end)], params)[0]
        self.__dict__.update(done.__dict__)

    def removeData(self):
        for name in list(self.__dict__.keys()):
            delattr(self, name)
