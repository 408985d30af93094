"""GPU: non-default parameters take the generic kernels (natac_background_generic, natac_occ_mle<0,0>, natac_occ_cov,
runtime-width smoothing): trimmed V-plot (W = 101, rows 110..240), occupancy with step 3 / flank 45 / upper 200,
smoothing sd 7, a 65-point alpha grid -- all against the oracle."""
import numpy as np
import pytest

from helpers import assert_track, expand_grid, golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("vlo,vup,w", [(110, 240, 50), (111, 240, 50), (107, 250, 60), (1, 60, 20)])
def test_generic_vmat_geometry(vlo, vup, w):
    """V-plots of other shapes: even / odd lower bound, odd number of rows (the FFT kernel pads a row pair), a width other than
    121, and a V-plot that includes insert size 1 (single-cell row: generic kernels)"""
    from nucleoatac_amd.device import Context
    from oracle import natac_oracle as O
    par = golden("params_example")
    if vlo >= 105:
        vm = np.ascontiguousarray(par["vmat"][vlo - 105:vup - 105, 60 - w:60 + w + 1])
    else:
        rng = np.random.default_rng(3)
        vm = rng.random((vup - vlo, 2 * w + 1)) * 0.01 + 1e-4
    sizes = synth_size_distribution(251)[:vup]
    pk = make_synthetic_chunks(40, 777, 260, seed=17)
    with Context(0) as c:
        c.set_vmat(vm, vlo, vup)
        c.set_sizes(sizes)
        b = c.upload(pk)
        b.run_nuc(7)
        tr = {t: b.split(b.track(t)) for t in (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH)}
        cc, cp, lr, var, z = b.run_peaks(min_signal=0, sep=21, boundary=40, order=10)
        for k in (0, 13, 39):
            l, n = pk.chunk_frags(k)
            nt = O.nuc_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, 777, pk.chunk_bias(k), -246, vm, vlo, vup, sizes,
                                    smooth_sd=7)
            assert_track(tr[L.T_NUC_COV][k], nt["nuc_cov"], "nuc_cov", exact=True)
            assert_track(tr[L.T_NFR_COV][k], nt["nfr_cov"], "nfr_cov", exact=True)
            assert_track(tr[L.T_RAW][k], nt["raw"], "raw")
            assert_track(tr[L.T_BACKGROUND][k], nt["bg"], "bg")
            assert_track(tr[L.T_NORM][k], nt["norm"], "norm")
            assert_track(tr[L.T_SMOOTH][k], nt["smoothed"], "smoothed")
            hp = O.call_peaks((tr[L.T_NORM][k] + tr[L.T_SMOOTH][k]).copy(), min_signal=0, sep=21, boundary=40, order=10)
            mine = cp[cc == k]
            assert np.array_equal(mine, hp)
            for pos, lrv, varv in list(zip(mine, lr[cc == k], var[cc == k]))[:4]:
                ref_lr = O.get_lr(nt["mat"], nt["mat_start"], nt["bmat"], nt["b0"], nt["b_start"], vm, vlo, vup, int(pos))
                if np.isnan(ref_lr):     # a zero model cell (zero size probability): NaN in the reference as well
                    assert np.isnan(lrv)
                else:
                    assert abs(lrv - ref_lr) <= 1e-7 * max(1.0, abs(ref_lr))
                pr = O.signal_distribution_probs(nt["bmat"], nt["b_start"], vlo, vup, w, int(pos))
                ref_var = O.calculate_cov_closed(pr, np.ravel(vm), nt["nuc_cov"][pos])
                assert abs(varv - ref_var) <= 1e-7 * max(1e-12, abs(ref_var))
        b.free()


def test_generic_occupancy_parameters():
    from nucleoatac_amd.device import Context
    from oracle import natac_oracle as O
    upper, flank, step = 200, 45, 3
    nucp, nfrp = synth_occ_distributions(251)
    nucp, nfrp = nucp[:upper] / nucp[:upper].sum(), nfrp[:upper] / nfrp[:upper].sum()
    alphas = np.linspace(0, 1, 65)
    pk = make_synthetic_chunks(30, 641, 220, seed=23)
    with Context(0) as c:
        c.set_occ_model(nucp, nfrp, alphas=alphas, cutoff=3.841458820694124, step=step, flank=flank)   # chi2.ppf(0.95, 1)
        b = c.upload(pk)
        b.run_occ()
        assert not b.status().any()
        grids = [b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)]
        sm = {t: b.split(b.track(t)) for t in (L.T_OCC_PREFILL, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV)}
        nk = len(range(1, 641, 3))
        for k in (0, 11, 29):
            l, n = pk.chunk_frags(k)
            oc = O.occ_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, 641, pk.chunk_bias(k), -246, nucp, nfrp, upper=upper,
                                    flank=flank, step=step, cutoff=3.841458820694124, n_alpha=65)
            for gi, key in enumerate(("occ", "occ_lower", "occ_upper")):
                assert_track(expand_grid(grids[gi][k * nk:(k + 1) * nk], 641, step), oc[key], key, exact=True)
            assert_track(sm[L.T_OCC_PREFILL][k], oc["smoothed_vals"], "smoothed")
            assert_track(sm[L.T_OCC_LOWER][k], oc["smoothed_lower"], "smoothed_lower")
            assert_track(sm[L.T_OCC_UPPER][k], oc["smoothed_upper"], "smoothed_upper")
            assert_track(sm[L.T_OCC_COV][k], oc["cov"], "cov", exact=True)
        b.free()
        # even step is decremented like the reference (Occupancy.py:190-191)
        c.set_occ_model(nucp, nfrp, alphas=alphas, cutoff=3.84, step=4, flank=flank)
        assert c.occ_step == 3


@pytest.mark.parametrize("step,flank,n_alpha,upper,zero", [
    (3, 60, 101, 251, None), (5, 61, 101, 251, None), (5, 62, 101, 251, None), (7, 62, 101, 251, None), (9, 44, 65, 200, None),
    (1, 20, 37, 251, None), (1, 61, 65, 237, None), (3, 120, 101, 251, None), (5, 60, 95, 251, None), (9, 4, 11, 251, None), (3, 73, 2, 251, None),
    (5, 60, 101, 251, "nfr"), (3, 61, 51, 251, "nfr"), (5, 60, 101, 251, "nuc"), (7, 33, 101, 251, "both sides")])
def test_fast_occupancy_path_for_any_odd_step_and_flank(step, flank, n_alpha, upper, zero):
    """--step 3 or --flank 61 used to drop the whole stage to the sliding-window kernel (5x slower).  The block-sum
    kernels now take any odd step up to 9, any flank (a window = whole step-blocks + the first (2 flank + 1) % step bases of the next
    one) and any increasing alpha grid of up to 101 values: alpha indices bit-exact against the oracle's literal log arithmetic and
    against the general kernel (NATAC_OCC_GENERAL=1); smoothed tracks (natac_occ_smooth_blk<STEP> for any flank) against the oracle and,
    bit for bit, against the one-base-per-lane smoothing kernel behind OCC_PREFILL"""
    import os
    from nucleoatac_amd.device import Context
    from oracle import natac_oracle as O
    nucp, nfrp = synth_occ_distributions(251)
    nucp, nfrp = nucp[:upper].copy(), nfrp[:upper].copy()
    # exact zeros in the models (Occupancy.py:112-114: 0 * log 0 = NaN -> -inf shuts out alpha = 0 / alpha = 1 for EVERY window; a
    # fragment whose size has nfr_prob 0 contributes the factor alpha): still the block-sum kernels, no size is 0 under both
    if zero in ("nfr", "both sides"):
        nfrp[170:] = 0.0
    if zero in ("nuc", "both sides"):
        nucp[:90] = 0.0
    nucp, nfrp = nucp / nucp.sum(), nfrp / nfrp.sum()
    alphas = np.linspace(0, 1, n_alpha)
    Lc = 1203
    counts = np.full(24, 330, dtype=np.int64)
    counts[7::8] = 14                       # sparse chunks: fragment-free stretches -> NaN blocks inside the smoothing windows
    pk = make_synthetic_chunks(24, Lc, 330, seed=step * 100 + flank, counts=counts)
    out = {}
    for mode in ("fast", "general"):
        if mode == "general":
            os.environ["NATAC_OCC_GENERAL"] = "1"
        try:
            with Context(0) as c:
                c.set_occ_model(nucp, nfrp, alphas=alphas, step=step, flank=flank)
                b = c.upload(pk)
                b.run_occ()
                assert not b.status().any()
                out[mode] = ([b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)],
                             [b.track(t) for t in (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV)], b.track(L.T_OCC_PREFILL))
                b.free()
        finally:
            os.environ.pop("NATAC_OCC_GENERAL", None)
    for a, g in zip(out["fast"][0] + out["fast"][1], out["general"][0] + out["general"][1]):
        assert np.array_equal(a, g, equal_nan=True)
    nk = len(range((step - 1) // 2, Lc, step))
    pre = out["fast"][2]
    occ, lo, hi, cov = out["fast"][1]
    for k in (0, 7, 23):
        l, n = pk.chunk_frags(k)
        oc = O.occ_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, Lc, pk.chunk_bias(k), -246, nucp, nfrp, upper=upper,
                                flank=flank, step=step, n_alpha=n_alpha)
        for gi, key in enumerate(("occ", "occ_lower", "occ_upper")):
            assert_track(expand_grid(out["fast"][0][gi][k * nk:(k + 1) * nk], Lc, step), oc[key], key, exact=True)
        sl = slice(k * Lc, (k + 1) * Lc)
        assert_track(pre[sl], oc["smoothed_vals"], "smoothed")
        assert_track(lo[sl], oc["smoothed_lower"], "smoothed_lower")
        assert_track(hi[sl], oc["smoothed_upper"], "smoothed_upper")
        assert_track(cov[sl], oc["cov"], "cov", exact=True)
        gap = np.isnan(pre[sl])
        assert np.array_equal(occ[sl][~gap], pre[sl][~gap])          # block-per-lane smoothing == one-base-per-lane smoothing, bit for bit


def test_heavy_tiles_first_does_not_show_in_any_result():
    """natac_occ_decide visits the tiles with more than max(256, 4 x mean) fragments first (natac_tile_heavy: a 10x denser chunk is a 10x
    longer wave).  The order of the launch must not show anywhere: a batch with a few 12x denser chunks, run with the heavy-first slots and
    with NATAC_OCC_ORDER=0 (chunk order), gives the same grid values, tracks and status words, bit for bit -- and matches the oracle on a
    dense and an ordinary chunk (the dense tiles' fragments do not fit the LDS strip: global-memory path of the decision kernel)."""
    import os
    from nucleoatac_amd.device import Context
    from oracle import natac_oracle as O
    nucp, nfrp = synth_occ_distributions(251)
    counts = np.full(60, 400, dtype=np.int64)
    counts[[3, 17, 18, 44]] = 5000
    counts[9] = 0
    pk = make_synthetic_chunks(60, 2120, 400, seed=77, counts=counts)
    out = {}
    for order in ("1", "0"):
        os.environ["NATAC_OCC_ORDER"] = order
        try:
            with Context(0) as c:
                c.set_occ_model(nucp, nfrp, step=5, flank=60)
                b = c.upload(pk)
                b.run_occ()
                out[order] = [b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)] + \
                             [b.track(t) for t in (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV)] + [b.status()]
                b.free()
        finally:
            os.environ.pop("NATAC_OCC_ORDER", None)
    for a, g in zip(out["1"], out["0"]):
        assert np.array_equal(a, g, equal_nan=a.dtype.kind == "f")
    assert not out["1"][-1].any()
    nk = len(range(2, 2120, 5))
    for k in (3, 4, 9):
        l, n = pk.chunk_frags(k)
        oc = O.occ_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, 2120, pk.chunk_bias(k), -246, nucp, nfrp)
        for gi, key in enumerate(("occ", "occ_lower", "occ_upper")):
            assert_track(expand_grid(out["1"][gi][k * nk:(k + 1) * nk], 2120, 5), oc[key], key, exact=True)
