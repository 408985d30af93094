"""Randomised sweep over the single-region drop-ins (the Cython functions' replacements and the operator-level helpers) against
the CPU oracle: makeFragmentMat, getInsertions, getFragmentSizesFromChunkList, calculateCov (closed + literal), smooth in
every mode, makeBiasMat, the PWM score, correlate 'valid', calculateOccupancy.   usage: python tests/fuzz/fuzz_dropins.py [rounds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from nucleoatac_amd.device import Context  # noqa: E402
from nucleoatac_amd.synth import synth_occ_distributions  # noqa: E402
from oracle import natac_oracle as O  # noqa: E402


def close(a, b, rtol=1e-9, atol=1e-12):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    assert np.all(np.abs(a[m] - b[m]) <= atol + rtol * np.abs(b[m])), float(np.max(np.abs(a[m] - b[m])))


def one_round(c, rng):
    # fragments around a region
    start = int(rng.integers(0, 5000))
    end = start + int(rng.integers(1, 3000))
    nf = int(rng.integers(0, 4000))
    l = rng.integers(start - 600, end + 600, size=nf).astype(np.int64)
    n = rng.integers(0, 700, size=nf).astype(np.int64)
    lower = int(rng.integers(0, 120))
    upper = lower + int(rng.integers(1, 300))
    assert np.array_equal(c.make_fragment_mat(l, n, start, end, lower, upper), O.make_fragment_mat(l, n, start, end, lower, upper))
    assert np.array_equal(c.get_insertions(l, n, start, end, lower, upper), O.get_insertions(l, n, start, end, lower, upper))
    # size histogram over a random (overlapping / empty / unsorted) chunk list
    nch = int(rng.integers(1, 40))
    cs = rng.integers(start - 800, end + 800, size=nch).astype(np.int64)
    ce = cs + rng.integers(0, 900, size=nch)
    got = c.fragment_sizes(l, n, cs, ce, lower, upper)
    assert np.array_equal(got, O.fragment_sizes_from_chunks(l, n, cs, ce, lower, upper))
    # multinomial variance
    m = int(rng.integers(2, 3000))
    p = rng.random(m) ** 3
    p /= p.sum()
    v = rng.random(m)
    r = int(rng.integers(1, 500))
    close(c.calculate_cov(p, v, r), O.calculate_cov_closed(p, v, r), rtol=1e-9)
    if m <= 600:
        close(c.calculate_cov(p, v, r, literal=True), O.calculate_cov_literal(p, v, r), rtol=1e-9)
    # smoothing, all modes / windows, with NaNs
    sig = rng.normal(0, 1, size=int(rng.integers(130, 4000)))
    if rng.random() < 0.5:
        sig[rng.integers(0, len(sig), size=len(sig) // 15)] = np.nan
    wl = int(rng.choice([3, 11, 61, 121]))
    for window in ("flat", "gaussian"):
        for mode in ("valid", "same"):
            for norm in (True, False):
                sd = float(rng.choice([2.0, 10.0, 20.0])) if window == "gaussian" else None
                close(c.smooth(sig, wl, window=window, sd=sd, mode=mode, norm=norm), O.smooth(sig, wl, window=window, sd=sd, mode=mode, norm=norm),
                      rtol=1e-9, atol=1e-11)
    # bias matrix + PWM score
    nbias = (end - start) + 700
    bl = rng.normal(0, 0.8, size=nbias)
    ts = start - 350
    lo2 = int(rng.integers(0, 100))
    up2 = lo2 + int(rng.integers(1, 250))
    s2 = start + int(rng.integers(0, max(1, (end - start) // 2)))
    e2 = min(end, s2 + int(rng.integers(1, 400)))
    close(c.make_bias_mat(bl, ts, s2, e2, lo2, up2), O.make_bias_mat(bl, ts, s2, e2, lo2, up2), rtol=1e-12)
    K = int(rng.choice([11, 21]))
    pwm = rng.random((4, K)) + 0.01
    pwm /= pwm.sum(axis=0)
    nuc = ["A", "C", "G", "T"]
    seq = "".join(rng.choice(list("ACGTN"), size=int(rng.integers(K, 2000)), p=[0.24, 0.24, 0.24, 0.24, 0.04]))
    close(c.pwm_bias(seq, pwm, nuc), O.compute_bias_pwm(seq, pwm, nuc), rtol=1e-12, atol=1e-12)
    # dense 'valid' correlation
    R, W = int(rng.integers(1, 150)), int(rng.choice([21, 61, 121]))
    vm = rng.random((R, W))
    sub = rng.random((R, W + int(rng.integers(0, 500))))
    close(c.correlate_valid(sub, vm), O.correlate_valid(sub, vm), rtol=1e-10, atol=1e-10)
    # single-window occupancy
    ins = rng.integers(0, 4, size=251).astype(np.float64) * (rng.random(251) < 0.2)
    bias = np.exp(rng.normal(0, 1, size=251))
    got = c.calculate_occupancy(ins, bias)
    if ins.sum() > 0:
        nucp, nfrp = one_round.model
        ref = O.calculate_occupancy(ins, bias, nucp, nfrp, np.linspace(0, 1, 101), 2.705543454095404)
        assert tuple(got) == tuple(ref), (got, ref)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    one_round.model = synth_occ_distributions(251)
    t0 = time.time()
    done = 0
    with Context(0) as c:
        c.set_occ_model(*one_round.model, step=5, flank=60)
        for _ in range(rounds):
            one_round(c, rng)
            done += 1
            if time.time() - t0 > float(os.environ.get("FUZZ_SECONDS", "1e9")):
                break
    print("drop-in fuzz ok: %d rounds, %.0f s" % (done, time.time() - t0))


if __name__ == "__main__":
    main()
