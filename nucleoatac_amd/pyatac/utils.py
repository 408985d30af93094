"""smoothing + peak calling helpers (API of the reference's pyatac/utils.py:23-134).

`smooth` runs on the GPU (natac_smooth); `call_peaks` / `reduce_peaks` are the host-side greedy peak logic that
sits between the signal kernels and the candidate kernel.
"""
import numpy as np


def smooth(sig, window_len, window="flat", sd=None, mode="valid", norm=True):
    """flat / gaussian smoothing with NaN-aware normalisation (pyatac/utils.py:23-52) -- natac_smooth on the GPU"""
    from .. import get_context
    return get_context().smooth(np.asarray(sig, dtype=np.float64), window_len, window=window, sd=sd, mode=mode, norm=norm)


def reduce_peaks(peaks, sig, sep):
    """greedy thinning: keep the strongest peak, drop everything closer than `sep`, repeat (pyatac/utils.py:56-78)"""
    peaks = np.asarray(peaks)
    n = peaks.size
    keep = np.zeros(n, dtype=bool)
    dead = np.zeros(n, dtype=bool)
    # stable order (equal heights: the later position first), the device kernel's rule; the reference's default argsort is an
    # unstable introsort, so among exactly equal heights its visiting order depends on the numpy build (utils.py:61)
    for ind in np.argsort(sig, kind="stable")[::-1]:
        if dead[ind]:
            continue
        keep[ind] = dead[ind] = True
        k = ind - 1
        while k >= 0 and peaks[ind] - peaks[k] < sep:
            dead[k] = True
            k -= 1
        k = ind + 1
        while k < n and peaks[k] - peaks[ind] < sep:
            dead[k] = True
            k += 1
    return peaks[keep]


def call_peaks(sigvals, min_signal=0, sep=120, boundary=None, order=1):
    """local maxima (with the reference's seeded 1e-12 jitter tie-break) thinned greedily (pyatac/utils.py:82-102).
    NaNs of `sigvals` are replaced IN PLACE by the minimum finite value, like the reference."""
    from scipy import signal
    nan = np.isnan(sigvals)
    if nan.any():
        if nan.all():
            return np.array([])
        sigvals[nan] = np.min(sigvals[~nan])
    if boundary is None:
        boundary = sep // 2
    n = len(sigvals)
    jitter = np.random.RandomState(seed=25).uniform(0, 10 ** -12, n)
    peaks = signal.argrelmax(sigvals * (1 + jitter), order=order)[0]
    peaks = peaks[sigvals[peaks] >= min_signal]
    peaks = peaks[(peaks >= boundary) & (peaks < n - boundary)]
    return reduce_peaks(peaks, sigvals[peaks], sep)


def read_chrom_sizes(sizesFile):
    """chrom sizes from a two-column file (.fai works) (pyatac/utils.py:126-134)"""
    out = {}
    with open(sizesFile) as f:
        for line in f:
            k = line.rstrip("\n").split("\t")
            if len(k) >= 2:
                out[k[0]] = int(k[1])
    return out


def read_chrom_sizes_from_fasta(fastafile):
    from .seq import FastaStore
    return FastaStore.sizes(fastafile)


def read_chrom_sizes_from_bam(bamfile):
    from .fragments import FragmentStore
    return FragmentStore.open(bamfile).chrom_sizes()
