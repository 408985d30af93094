"""CPU restatement (numpy) of NucleoATAC's per-chunk occ + nuc signal path.

TEST INFRASTRUCTURE ONLY -- this module is the parity *checker* and the
`cpu_baseline` leg of bench.py.  Nothing in `nucleoatac_amd/` (the product)
imports it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may.  It is deliberately written in the reference's execution shape (one
chunk at a time, dense numpy matrices, scipy/numpy per-stage calls) so that it
can stand in for the reference's CPU path when timing.

Parity pin: every function below was checked in the build container against a
Python-3 scratch copy of the reference itself (oracle/make_scratch_ref.py) on
seeded synthetic chunks and on the reference's own fixtures; the resulting
input/output vectors are committed under tests/golden/ (generator:
tests/golden/make_golden.py) and tests/test_oracle_golden.py re-checks this
module against them on every run.

All `file:line` citations are relative to the reference repository
(GreenleafLab/NucleoATAC v0.3.4).  A "fragment" is one forward-strand
proper-pair read already converted to (l, n): l = pos+4 (left Tn5 insertion),
n = |tlen|-8 (insertion-to-insertion length)  [pyatac/fragments.pyx:26-31].
`//` is Python floor division (the reference is Python 2: int/int floors).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------------------
# fragments  (pyatac/fragments.pyx)
# --------------------------------------------------------------------------------------


def fragment_center(l, n):
    """centre column of a fragment: l + (n-1)//2   [pyatac/fragments.pyx:36]"""
    return l + (n - 1) // 2


def make_fragment_mat(l, n, start, end, lower, upper):
    """V-plot count matrix (rows = insert size, cols = fragment centre).

    Restates makeFragmentMat [pyatac/fragments.pyx:17-40]: row = n-lower,
    col = (n-1)//2 + l - start, +1 when both are in bounds.  float64 like the
    reference (counts are integer valued).
    """
    nrow, ncol = upper - lower, end - start
    mat = np.zeros((nrow, ncol), dtype=np.float64)
    row = n - lower
    col = (n - 1) // 2 + l - start
    ok = (col >= 0) & (col < ncol) & (row >= 0) & (row < nrow)
    np.add.at(mat, (row[ok], col[ok]), 1.0)
    return mat


def get_insertions(l, n, start, end, lower=0, upper=2000):
    """per-base Tn5 insertion counts  [pyatac/fragments.pyx:43-67].

    For lower <= n < upper: +1 at l and +1 at r = l+n-1 when inside [start,end).
    """
    out = np.zeros(end - start, dtype=np.float64)
    keep = (n >= lower) & (n < upper)
    lk, rk = l[keep], l[keep] + n[keep] - 1
    a = lk[(lk >= start) & (lk < end)] - start
    b = rk[(rk >= start) & (rk < end)] - start
    np.add.at(out, a, 1.0)
    np.add.at(out, b, 1.0)
    return out


def get_stranded_insertions(l, n, start, end, lower=0, upper=2000):
    """(plus, minus) per-base insertion counts  [pyatac/fragments.pyx:71-97]: for lower <= n < upper the left end l
    counts on the plus strand, the right end r = l+n-1 on the minus strand, each when inside [start,end)."""
    plus = np.zeros(end - start, dtype=np.float64)
    minus = np.zeros(end - start, dtype=np.float64)
    keep = (n >= lower) & (n < upper)
    lk, rk = l[keep], l[keep] + n[keep] - 1
    np.add.at(plus, lk[(lk >= start) & (lk < end)] - start, 1.0)
    np.add.at(minus, rk[(rk >= start) & (rk < end)] - start, 1.0)
    return plus, minus


def get_ins_from_mat(mat, lower, upper):
    """ChunkMat2D.getIns [pyatac/chunkmat2d.py:74-84]: collapse a V-plot matrix to insertions.

    Equivalent to correlate2d(mat, pattern, 'valid')[0] where pattern row i has ones at
    mid+(i-1)//2 and mid-(i//2), mid = upper//2.  Output spans [start+P//2, end-P//2),
    P = upper + (upper-1)%2.
    """
    P = upper + (upper - 1) % 2
    mid = upper // 2
    ncol = mat.shape[1]
    nout = ncol - P + 1
    out = np.zeros(nout, dtype=np.float64)
    for i in range(lower, upper):
        row = mat[i - lower]
        cols = {mid + (i - 1) // 2, mid - (i // 2)}  # a single cell when i == 1 (and i == 0)
        for c in cols:
            out += row[c:c + nout]
    return out, P // 2


def fragment_sizes_from_chunks(l, n, chunk_starts, chunk_ends, lower, upper):
    """getFragmentSizesFromChunkList [pyatac/fragments.pyx:123-145] on one chromosome's fragments.

    A fragment is counted once per chunk that contains its centre.
    """
    sizes = np.zeros(upper - lower, dtype=np.float64)
    c = fragment_center(l, n)
    ok = (n >= lower) & (n < upper)
    for s, e in zip(chunk_starts, chunk_ends):
        m = ok & (c >= s) & (c < e)
        np.add.at(sizes, n[m] - lower, 1.0)
    return sizes


def normalise_sizes(sizes):
    """FragmentSizes.calculateSizes normalisation [pyatac/fragmentsizes.py:27]"""
    t = np.sum(sizes)
    return sizes / (t + (t == 0))


# --------------------------------------------------------------------------------------
# Tn5 bias  (pyatac/bias.py, pyatac/seq.py, pyatac/chunkmat2d.py)
# --------------------------------------------------------------------------------------


def compute_bias_pwm(sequence, pwm_mat, nucleotides):
    """InsertionBiasTrack.computeBias [pyatac/bias.py:85-92] + seq_to_mat [pyatac/seq.py:37-45].

    `sequence` covers [start-up, end+down); returns log-bias for [start, end):
    b[x] = sum_k log PWM[nuc(seq[x+k]), k]; characters not in `nucleotides` add 0.
    """
    logp = np.log(pwm_mat)
    K = pwm_mat.shape[1]
    nout = len(sequence) - K + 1
    seq = np.frombuffer(sequence.encode("ascii"), dtype=np.uint8)
    out = np.zeros(nout, dtype=np.float64)
    for r, nuc in enumerate(nucleotides):
        hit = (seq == ord(nuc)).astype(np.float64)
        out += np.correlate(hit, logp[r], mode="valid")
    return out


def make_bias_mat(bias_log, track_start, start, end, lower, upper):
    """BiasMat2D.makeBiasMat [pyatac/chunkmat2d.py:140-153] for a log-scale bias track.

    B0[i-lower, x] = exp(b[g - (i-1)//2] + b[g + i//2]),  g = start + x  -- except insert
    size i == 1, where the reference's two pattern ones coincide and B0 = exp(b[g]).
    `bias_log` is the track, `track_start` its first genomic coordinate.
    """
    ncol = end - start
    mat = np.empty((upper - lower, ncol), dtype=np.float64)
    g = np.arange(start, end) - track_start
    for i in range(lower, upper):
        if (i - 1) // 2 == -(i // 2):  # i == 1: both pattern ones land on the same cell (:150-151)
            mat[i - lower] = np.exp(bias_log[g])
        else:
            mat[i - lower] = np.exp(bias_log[g - (i - 1) // 2] + bias_log[g + i // 2])
    return mat


def norm_by_insert_dist(bias_mat, sizes, lower, upper, sizes_lower=0):
    """BiasMat2D.normByInsertDist [pyatac/chunkmat2d.py:154-156]: row i scaled by sizes[i]."""
    return bias_mat * sizes[lower - sizes_lower:upper - sizes_lower][:, None]


# --------------------------------------------------------------------------------------
# smoothing / coverage  (pyatac/utils.py, pyatac/tracks.py)
# --------------------------------------------------------------------------------------


def gaussian_window(M, sd):
    """scipy.signal.gaussian(M, sd): exp(-0.5 (n/sd)^2), n = arange(M) - (M-1)/2 [pyatac/utils.py:40]."""
    nn = np.arange(M) - (M - 1) / 2.0
    return np.exp(-0.5 * (nn / sd) ** 2)


def smooth(sig, window_len, window="flat", sd=None, mode="valid", norm=True):
    """utils.smooth [pyatac/utils.py:23-52] (NaN-aware when norm=True)."""
    if window_len % 2 != 1:
        window_len += 1
    if window == "gaussian":
        if sd is None:
            sd = (window_len - 1) / 6.0
        w = gaussian_window(window_len, sd)
    else:
        w = np.ones(window_len)
    nan = np.isnan(sig)
    s0 = np.where(nan, 0.0, sig)
    sm = np.convolve(w, s0, mode=mode)
    if norm:
        nrm = np.convolve(w, (~nan).astype(np.float64), mode=mode)
        nrm[nrm == 0] = np.nan
        sm = sm / nrm
    return sm


def coverage(mat, mat_start, mat_lower, start, end, lower, upper, window_len):
    """CoverageTrack.calculateCoverage [pyatac/tracks.py:209-222].

    cov[g] = sum_{rows lower..upper} sum_{|d| <= window_len//2} mat[row, g+d], g in [start,end).
    """
    offset = start - mat_start - window_len // 2
    assert offset >= 0
    sub = mat[lower - mat_lower:upper - mat_lower, offset:mat.shape[1] - offset if offset else None]
    collapsed = np.sum(sub, axis=0)
    return smooth(collapsed, window_len, window="flat", mode="valid", norm=False)


# --------------------------------------------------------------------------------------
# NucleoATAC signal  (nucleoatac/NucleosomeCalling.py)
# --------------------------------------------------------------------------------------


def correlate_valid(sub, vmat):
    """signal.correlate(sub, vmat, 'valid')[0] as an explicit sum of per-row 1-D correlations
    [nucleoatac/NucleosomeCalling.py:34-36, 60-63]."""
    out = np.zeros(sub.shape[1] - vmat.shape[1] + 1, dtype=np.float64)
    for r in range(vmat.shape[0]):
        out += np.correlate(sub[r], vmat[r], mode="valid")
    return out


def nuc_chunk_tracks(l, n, start, end, bias_log, bias_track_start, vmat, vlower, vupper, sizes, smooth_sd=10,
                     dense_correlate=correlate_valid):
    """NucChunk.process up to smoothSignal [nucleoatac/NucleosomeCalling.py:239-283, 328-334].

    vmat: (vupper-vlower, 2w+1); sizes: global insert-size distribution over [0, vupper).
    bias_log None == fasta None (bias matrix of ones, NucleosomeCalling.py:243-250).
    Returns dict of per-base tracks over [start, end) plus the dense matrices.
    """
    W = vmat.shape[1]
    w = W // 2
    upper = vupper
    flank_m = max(W, upper // 2 + 1)  # :240
    mat_start, mat_end = start - flank_m, end + flank_m
    mat = make_fragment_mat(l, n, mat_start, mat_end, 0, upper)
    b_start, b_end = start - W, end + W  # :244
    if bias_log is not None:
        b0 = make_bias_mat(bias_log, bias_track_start, b_start, b_end, 0, upper)
    else:
        b0 = np.ones((upper, b_end - b_start))
    bmat = norm_by_insert_dist(b0, sizes, 0, upper)
    nuc_cov = coverage(mat, mat_start, 0, start, end, vlower, vupper, W)  # :257-260
    nfr_cov = coverage(mat, mat_start, 0, start, end, 0, vlower, W)  # :271-273
    # BiasTrack.calculateBackgroundSignal :49-64
    off_b = start - b_start - w
    sub_b = bmat[vlower:vupper, off_b:bmat.shape[1] - off_b]
    cov_b = coverage(bmat, b_start, 0, start, end, vlower, vupper, W)
    bg_num = dense_correlate(sub_b, vmat)
    with np.errstate(invalid="ignore", divide="ignore"):
        bg = bg_num * nuc_cov / cov_b
    # SignalTrack.calculateSignal :29-36
    off_m = start - mat_start - w
    sub_m = mat[vlower:vupper, off_m:mat.shape[1] - off_m]
    raw = dense_correlate(sub_m, vmat)
    norm = raw - bg  # :38-43
    # smoothSignal :274-283
    tmp = norm.copy()
    tmp[tmp < 0] = 0
    smoothed = smooth(tmp, 6 * smooth_sd + 1, window="gaussian", sd=smooth_sd, mode="same", norm=True)
    return dict(nuc_cov=nuc_cov, nfr_cov=nfr_cov, raw=raw, bg=bg, norm=norm, smoothed=smoothed,
                bg_num=bg_num, cov_b=cov_b, mat=mat, mat_start=mat_start, b0=b0, bmat=bmat, b_start=b_start)


def get_lr(mat, mat_start, bmat, b0, b_start, vmat, vlower, vupper, pos):
    """Nucleosome.getLR [nucleoatac/NucleosomeCalling.py:110-122]."""
    w = vmat.shape[1] // 2
    m = mat[vlower:vupper, pos - w - mat_start:pos + w + 1 - mat_start]
    null_mat = bmat[vlower:vupper, pos - w - b_start:pos + w + 1 - b_start]
    bias_mat = b0[vlower:vupper, pos - w - b_start:pos + w + 1 - b_start]
    nuc_model = vmat * bias_mat
    nuc_model = nuc_model / np.sum(nuc_model)
    null_model = null_mat / np.sum(null_mat)
    with np.errstate(invalid="ignore", divide="ignore"):
        nuc_lik = np.sum(np.log(nuc_model) * m)
        null_lik = np.sum(np.log(null_model) * m)
    return nuc_lik - null_lik


_covlib = None


def _load_covlib():
    global _covlib
    if _covlib is None:
        so = os.path.join(_HERE, "_build", "libnatac_oracle.so")
        if not os.path.exists(so):
            from oracle.build_oracle import build_oracle_c
            build_oracle_c()
        _covlib = ctypes.CDLL(so)
        _covlib.oracle_calculate_cov.restype = ctypes.c_double
        _covlib.oracle_calculate_cov.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]
    return _covlib


def calculate_cov_literal(p, v, r):
    """calculateCov [nucleoatac/multinomial_cov.pyx:20-31], literal O(N^2) double loop in the
    reference's summation order, accumulator = 0 (the .pyx leaves it uninitialised, :23),
    `r` truncated to C int (:20).  Implemented in oracle/natac_oracle_c.c."""
    p = np.ascontiguousarray(p, dtype=np.float64)
    v = np.ascontiguousarray(v, dtype=np.float64)
    if p.shape[0] != v.shape[0]:
        raise ValueError("p and v must be same shape")
    lib = _load_covlib()
    return lib.oracle_calculate_cov(p.ctypes.data, v.ctypes.data, p.shape[0], int(r))


def calculate_cov_closed(p, v, r):
    """closed form of the same quantity: r * (sum p v^2 - (sum p v)^2)
    (identity pinned by the reference's tests/test_var.py:34-43)."""
    return int(r) * (np.sum(p * v * v) - np.sum(p * v) ** 2)


def signal_distribution_probs(bmat, b_start, vlower, vupper, w, pos):
    """SignalDistribution.__init__ [nucleoatac/NucleosomeCalling.py:70-76]."""
    sub = bmat[vlower:vupper, pos - w - b_start:pos + w + 1 - b_start]
    return (sub / np.sum(sub)).flatten()


def z_score(norm_at_pos, nuc_cov_at_pos, probs, vmat, literal=False):
    """Nucleosome.getZScore [nucleoatac/NucleosomeCalling.py:123-127] / analStd :83-86."""
    v = np.ravel(vmat)
    f = calculate_cov_literal if literal else calculate_cov_closed
    var = f(probs, v, nuc_cov_at_pos)
    with np.errstate(invalid="ignore", divide="ignore"):
        return norm_at_pos / np.sqrt(var), var


# --------------------------------------------------------------------------------------
# Occupancy  (nucleoatac/Occupancy.py)
# --------------------------------------------------------------------------------------

CHI2_90_DF1 = 2.705543454095404  # scipy.stats.chi2.ppf(0.9, 1)  [nucleoatac/Occupancy.py:102]


def calculate_occupancy(inserts, bias, nuc_probs0, nfr_probs0, alphas, cutoff):
    """calculateOccupancy [nucleoatac/Occupancy.py:104-120]."""
    with np.errstate(invalid="ignore", divide="ignore"):
        nuc_probs = nuc_probs0 * bias
        nuc_probs = nuc_probs / np.sum(nuc_probs)
        nfr_probs = nfr_probs0 * bias
        nfr_probs = nfr_probs / np.sum(nfr_probs)
        logliks = np.array([np.sum(np.log(a * nuc_probs + (1 - a) * nfr_probs) * inserts) for a in alphas])
    logliks[np.isnan(logliks)] = -float("inf")
    occ = alphas[np.argmax(logliks)]
    with np.errstate(invalid="ignore"):
        ratios = 2 * (max(logliks) - logliks)
    idx = np.where(ratios < cutoff)[0]
    return occ, alphas[min(idx)], alphas[max(idx)]


def occ_chunk_tracks(l, n, start, end, bias_log, bias_track_start, nuc_probs, nfr_probs, upper=251, flank=60,
                     step=5, cutoff=CHI2_90_DF1, n_alpha=101):
    """OccChunk.process up to getCov [nucleoatac/Occupancy.py:204-224, 241-247].

    Returns occ/lower/upper (raw, piecewise constant with NaN gaps), their NaN-aware Gaussian
    smoothings (makeSmoothed :147-153, window 2*flank+1, sd flank/3.0) and cov.
    """
    if step % 2 == 0:
        step -= 1  # :190-191
    halfstep = (step - 1) // 2
    alphas = np.linspace(0, 1, n_alpha)
    mat_start, mat_end = start - flank, end + flank
    mat = make_fragment_mat(l, n, mat_start, mat_end, 0, upper)
    if bias_log is not None:
        b0 = make_bias_mat(bias_log, bias_track_start, mat_start, mat_end, 0, upper)
    else:
        b0 = np.ones((upper, mat_end - mat_start))
    L = end - start
    vals = np.full(L, np.nan)
    lo = np.full(L, np.nan)
    hi = np.full(L, np.nan)
    for i in range(halfstep, L, step):  # :136
        x0 = i  # start + i - flank - mat_start == i
        ins = np.sum(mat[:, x0:x0 + 2 * flank + 1], axis=1)
        bias = np.sum(b0[:, x0:x0 + 2 * flank + 1], axis=1)
        if np.sum(ins) > 0:
            a, b = i - halfstep, min(i + halfstep + 1, L)
            vals[a:b], lo[a:b], hi[a:b] = calculate_occupancy(ins, bias, nuc_probs, nfr_probs, alphas, cutoff)
    window = 2 * flank + 1
    sd = flank / 3.0
    sm = [smooth(x, window, window="gaussian", sd=sd, mode="same", norm=True) for x in (vals, lo, hi)]
    cov = coverage(mat, mat_start, 0, start, end, 0, upper, window)
    return dict(occ=vals, occ_lower=lo, occ_upper=hi, smoothed_vals=sm[0], smoothed_lower=sm[1],
                smoothed_upper=sm[2], cov=cov, mat=mat, mat_start=mat_start, b0=b0)


# --------------------------------------------------------------------------------------
# peak calling  (pyatac/utils.py)  -- host-side logic adjacent to the hot path
# --------------------------------------------------------------------------------------


def reduce_peaks(peaks, sig, sep):
    """utils.reduce_peaks [pyatac/utils.py:56-78]."""
    peaks = np.asarray(peaks)
    exclude = np.zeros(peaks.size)
    keep = np.zeros(peaks.size)
    st = np.argsort(sig)
    j = peaks.size - 1
    while j >= 0:
        ind = st[j]
        j -= 1
        if exclude[ind] == 0:
            keep[ind] = 1
            exclude[ind] = 1
            k = ind - 1
            while k >= 0 and (peaks[ind] - peaks[k]) < sep:
                exclude[k] = 1
                k -= 1
            k = ind + 1
            while k < peaks.size and (peaks[k] - peaks[ind]) < sep:
                exclude[k] = 1
                k += 1
    return peaks[keep == 1]


def call_peaks(sigvals, min_signal=0, sep=120, boundary=None, order=1):
    """utils.call_peaks [pyatac/utils.py:82-102]; fills NaNs of `sigvals` IN PLACE like the reference."""
    from scipy import signal
    nnan = int(np.sum(np.isnan(sigvals)))
    if nnan > 0:
        if nnan == len(sigvals):
            return np.array([])
        sigvals[np.isnan(sigvals)] = np.min(sigvals[~np.isnan(sigvals)])
    if boundary is None:
        boundary = sep // 2
    random = np.random.RandomState(seed=25)
    ln = len(sigvals)
    peaks = signal.argrelmax(sigvals * (1 + random.uniform(0, 10 ** -12, ln)), order=order)[0]
    peaks = peaks[sigvals[peaks] >= min_signal]
    peaks = peaks[peaks >= boundary]
    peaks = peaks[peaks < (ln - boundary)]
    return reduce_peaks(peaks, sigvals[peaks], sep)
