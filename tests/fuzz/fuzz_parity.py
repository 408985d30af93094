"""Randomised parity sweep on the GPU (development / release check): random ragged chunk sets -- lengths 121..9000, fragment
densities 0..12 per base, fragment-free stretches (NaN occupancy), with / without a bias track, odd sizes -- every track, the
occupancy grid, insertion counts and the candidate search against the CPU oracle.
usage: python tests/fuzz/fuzz_parity.py [n_rounds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_track, cancel_scale, golden  # noqa: E402
from nucleoatac_amd import _lib as L  # noqa: E402
from nucleoatac_amd.device import Context  # noqa: E402
from nucleoatac_amd.packing import PackedChunks  # noqa: E402
from nucleoatac_amd.synth import synth_occ_distributions, synth_size_distribution  # noqa: E402
from oracle import natac_oracle as O  # noqa: E402


def call_peaks_stable(sigvals, **kw):
    """O.call_peaks with reduce_peaks' argsort made STABLE.  The reference sorts with numpy's default introsort
    (pyatac/utils.py:61), so among peaks of exactly equal height -- occupancy saturated at 1.0 over a dense stretch -- its
    visiting order depends on the numpy build; the device uses the stable order (ties: the later position first), which
    is what the unstable sort returns whenever it does not permute equal keys."""
    import unittest.mock as mock
    real = np.argsort
    with mock.patch.object(np, "argsort", lambda a, *x, **k: real(a, kind="stable")):
        return O.call_peaks(sigvals, **kw)


def one_round(ctx, rng, par, sizes, nucp, nfrp, rnd):
    if rng.random() < 0.12:          # many short chunks: tile tables / offsets across chunk borders
        nch = int(rng.integers(40, 160))
        lens = [int(rng.integers(121, 520)) for _ in range(nch)]
    else:
        nch = int(rng.integers(1, 7))
        lens = [int(rng.choice([121, 122, 125, 126, int(rng.integers(127, 700)), int(rng.integers(700, 3000)),
                                int(rng.integers(3000, 9000))])) for _ in range(nch)]
    with_bias = rng.random() < 0.7
    fr = []
    for Lc in lens:
        dens = float(rng.choice([0.0, 0.02, 0.2, 1.0, 4.0, 12.0]))
        nf = int(dens * Lc) if Lc < 3000 or dens <= 1.0 else int(1.0 * Lc)
        n = rng.integers(1, 420, size=nf)
        if nf and rng.random() < 0.3:
            n[rng.integers(0, nf, size=max(1, nf // 20))] = rng.choice([0, 1, 2, 250, 251, 1999, 2500], size=max(1, nf // 20))
        c = rng.integers(-200, Lc + 200, size=nf)
        if nf and rng.random() < 0.5:            # a fragment-free stretch: NaN occupancy, all-NaN windows
            a = int(rng.integers(0, Lc))
            w = int(rng.integers(50, 600))
            keep = (c < a - 60) | (c > a + w + 60)
            c, n = c[keep], n[keep]
        o = np.argsort(c, kind="stable")
        c, n = c[o], n[o]
        fr.append((c - (n - 1) // 2, n))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    if with_bias:
        nb = [Lc + 493 for Lc in lens]
        bias = rng.normal(0, float(rng.choice([0.3, 0.8, 1.5])), size=sum(nb))
        pk = PackedChunks(np.arange(nch) * 20000, lens, off, np.concatenate([x[0] for x in fr]), np.concatenate([x[1] for x in fr]),
                          np.concatenate(([0], np.cumsum(nb))), bias)
    else:
        pk = PackedChunks(np.arange(nch) * 20000, lens, off, np.concatenate([x[0] for x in fr]), np.concatenate([x[1] for x in fr]),
                          None, None)
    b = ctx.upload(pk)
    b.run_nuc(10)
    b.run_occ()
    b.run_ins(0, 2000)
    st = b.status()
    tr = {t: b.split(b.track(t)) for t in (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH, L.T_OCC_PREFILL,
                                           L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV, L.T_INS)}
    kw = dict(min_signal=0, sep=int(rng.choice([25, 40, 120])), boundary=int(rng.choice([0, 30, 60])), order=int(rng.choice([1, 5, 12])))
    cc, cp, lr, var, z = b.run_peaks(**kw)
    min_occ = float(rng.choice([0.0, 0.1, 0.3]))
    osep = int(rng.choice([60, 120]))
    oc_c, oc_p, oc_occ, oc_lo, oc_up, oc_rd, oc_keep, nd = b.run_occ_peaks(min_occ=min_occ, sep=osep)
    st2 = b.status()
    for k, Lc in enumerate(lens):
        l, n = fr[k]
        bias_k = pk.chunk_bias(k) if with_bias else None
        nt = O.nuc_chunk_tracks(l, n, 0, Lc, bias_k, -246, par["vmat"], 105, 251, sizes)
        assert_track(tr[L.T_NUC_COV][k], nt["nuc_cov"], "nuc_cov", exact=True)
        assert_track(tr[L.T_NFR_COV][k], nt["nfr_cov"], "nfr_cov", exact=True)
        assert_track(tr[L.T_RAW][k], nt["raw"], "raw")
        assert_track(tr[L.T_BACKGROUND][k], nt["bg"], "bg")
        assert_track(tr[L.T_NORM][k], nt["norm"], "norm", scale=cancel_scale(nt["raw"], nt["bg"]))
        assert_track(tr[L.T_SMOOTH][k], nt["smoothed"], "smoothed", scale=cancel_scale(nt["raw"], nt["bg"]))
        assert np.array_equal(tr[L.T_INS][k], O.get_insertions(l, n, 0, Lc).astype(np.int32))
        if not (st[k] & 1):
            oc = O.occ_chunk_tracks(l, n, 0, Lc, bias_k, -246, nucp, nfrp)
            assert_track(tr[L.T_OCC_PREFILL][k], oc["smoothed_vals"], "occ")
            assert_track(tr[L.T_OCC_LOWER][k], oc["smoothed_lower"], "occ lower")
            assert_track(tr[L.T_OCC_UPPER][k], oc["smoothed_upper"], "occ upper")
            assert_track(tr[L.T_OCC_COV][k], oc["cov"], "occ cov", exact=True)
            filled = oc["smoothed_vals"].copy()
            O.call_peaks(filled)
            assert_track(tr[L.T_OCC][k], filled, "occ post-fill")
        if not (st[k] & 1) and not (st2[k] & 2):
            # OccChunk.callPeaks + getNucDist (Occupancy.py:225-240) restated on the device's own tracks
            occv, lov, upv, covv = tr[L.T_OCC][k], tr[L.T_OCC_LOWER][k], tr[L.T_OCC_UPPER][k], tr[L.T_OCC_COV][k]
            pk_ = np.asarray(call_peaks_stable(occv.copy(), sep=osep, min_signal=min_occ), np.int64)
            m = oc_c == k
            if not np.array_equal(oc_p[m], pk_):
                raise AssertionError(("occ peaks", k, Lc, min_occ, osep, oc_p[m][:12], pk_[:12], int(np.isnan(occv).sum()), st[k], st2[k]))
            if not (np.array_equal(oc_occ[m], occv[pk_]) and np.array_equal(oc_lo[m], lov[pk_], equal_nan=True) and np.array_equal(oc_rd[m], covv[pk_])):
                raise AssertionError(("occ peak values", k, Lc, oc_occ[m][:6], occv[pk_][:6], oc_lo[m][:6], lov[pk_][:6], oc_rd[m][:6], covv[pk_][:6]))
            keep = (lov[pk_] > min_occ) & (covv[pk_] > 0)
            assert np.array_equal(oc_keep[m].astype(bool), keep)
            cen = l + (n - 1) // 2
            ref_nd = np.zeros(251)
            for p_ in pk_[keep]:
                sel = (cen >= p_ - 60) & (cen <= p_ + 60) & (n >= 0) & (n < 251)
                h = np.bincount(n[sel], minlength=251)[:251].astype(np.float64)
                ref_nd += h / h.sum()
            assert np.allclose(nd[k], ref_nd, rtol=1e-12, atol=1e-15), ("nuc_dist", k)
        if not (st[k] & 2):
            comb = tr[L.T_NORM][k] + tr[L.T_SMOOTH][k]
            hp = np.asarray(call_peaks_stable(comb.copy(), **kw), np.int64)
            assert np.array_equal(cp[cc == k], hp), ("peaks", k, Lc, kw)
    b.free()
    return sum(lens), int(off[-1])


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    ctx = Context(0)
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(sizes)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    rng = np.random.default_rng(seed)
    t0 = time.time()
    bp = fr = 0
    for r in range(rounds):
        a, f = one_round(ctx, rng, par, sizes, nucp, nfrp, r)
        bp += a
        fr += f
        if time.time() - t0 > float(os.environ.get("FUZZ_SECONDS", "1e9")):
            rounds = r + 1
            break
    ctx.close()
    print("fuzz ok: %d rounds, %d bases, %d fragments, %.0f s" % (rounds, bp, fr, time.time() - t0))


if __name__ == "__main__":
    main()
