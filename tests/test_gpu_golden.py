"""GPU parity against the golden vectors produced by the reference itself (tests/golden/make_golden.py).
Everything here goes through the C-ABI (libnatac_hip.so) on a real MI355X."""
import numpy as np
import pytest

from helpers import assert_track, expand_grid, golden, packed_from_golden
from nucleoatac_amd import _lib as L

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nucleoatac_amd.device import Context
    c = Context(0)
    p = golden("params_example")
    c.set_vmat(p["vmat"], int(p["vlower"]), int(p["vupper"]))
    c.set_sizes(p["sizes"])
    c.set_occ_model(p["nuc_probs"], p["nfr_probs"], p["alphas"], float(p["cutoff"]), step=5, flank=60)
    yield c
    c.close()


@pytest.mark.parametrize("case,with_bias", [("chunks_basic", True), ("chunks_gaps", True), ("chunks_nobias", False)])
def test_nuc_tracks_match_reference(ctx, case, with_bias):
    g = golden(case)
    pk = packed_from_golden(g, with_bias)
    b = ctx.upload(pk)
    b.run_nuc(smooth_sd=10)
    tracks = {t: b.split(b.track(t)) for t in (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH)}
    for k in range(pk.n_chunks):
        assert_track(tracks[L.T_NUC_COV][k], g["c%d_nuc_cov" % k], "nuc_cov", exact=True)
        assert_track(tracks[L.T_NFR_COV][k], g["c%d_nfr_cov" % k], "nfr_cov", exact=True)
        assert_track(tracks[L.T_RAW][k], g["c%d_raw" % k], "raw")
        assert_track(tracks[L.T_BACKGROUND][k], g["c%d_bg" % k], "background")
        assert_track(tracks[L.T_NORM][k], g["c%d_norm" % k], "norm")
        assert_track(tracks[L.T_SMOOTH][k], g["c%d_smoothed" % k], "smoothed")
    b.free()


@pytest.mark.parametrize("case,with_bias", [("chunks_basic", True), ("chunks_gaps", True), ("chunks_nobias", False)])
def test_occ_tracks_match_reference(ctx, case, with_bias):
    g = golden(case)
    pk = packed_from_golden(g, with_bias)
    b = ctx.upload(pk)
    b.run_occ()
    assert not b.status().any()
    grids = [b.grid(w) for w in (L.G_OCC, L.G_LOWER, L.G_UPPER)]
    _, total_grid, _ = b.grid_info()
    off = 0
    sm = {t: b.split(b.track(t)) for t in (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV, L.T_OCC_PREFILL)}
    for k in range(pk.n_chunks):
        Lk = int(pk.chunk_len[k])
        nk = len(range(2, Lk, 5))
        # the alpha grid values are discrete: bit-exact
        assert_track(expand_grid(grids[0][off:off + nk], Lk), g["c%d_occ" % k], "occ.vals", exact=True)
        assert_track(expand_grid(grids[1][off:off + nk], Lk), g["c%d_occ_lower" % k], "occ.lower_bound", exact=True)
        assert_track(expand_grid(grids[2][off:off + nk], Lk), g["c%d_occ_upper" % k], "occ.upper_bound", exact=True)
        off += nk
        assert_track(sm[L.T_OCC_PREFILL][k], g["c%d_occ_smoothed_prefill" % k], "smoothed_vals (pre-fill)")
        assert_track(sm[L.T_OCC][k], g["c%d_occ_smoothed" % k], "smoothed_vals")
        assert_track(sm[L.T_OCC_LOWER][k], g["c%d_occ_smoothed_lower" % k], "smoothed_lower")
        assert_track(sm[L.T_OCC_UPPER][k], g["c%d_occ_smoothed_upper" % k], "smoothed_upper")
        assert_track(sm[L.T_OCC_COV][k], g["c%d_occ_cov" % k], "occ cov", exact=True)
    assert off == total_grid
    b.free()


@pytest.mark.parametrize("case,with_bias", [("chunks_basic", True), ("chunks_gaps", True), ("chunks_nobias", False)])
def test_insertions_bit_exact(ctx, case, with_bias):
    g = golden(case)
    pk = packed_from_golden(g, with_bias)
    b = ctx.upload(pk)
    for lo, up, key in ((0, 2000, "ins2000"), (0, 251, "ins251")):
        b.run_ins(lo, up)
        ins = b.split(b.track(L.T_INS))
        for k in range(pk.n_chunks):
            assert ins[k].dtype == np.int32
            assert np.array_equal(ins[k], g["c%d_%s" % (k, key)].astype(np.int32)), (case, k, key)
    b.free()


@pytest.mark.parametrize("case,with_bias", [("chunks_basic", True), ("chunks_gaps", True), ("chunks_nobias", False)])
def test_candidate_lr_var_z(ctx, case, with_bias):
    g = golden(case)
    pk = packed_from_golden(g, with_bias)
    b = ctx.upload(pk)
    b.run_nuc(smooth_sd=10)
    cc, cp, ref = [], [], []
    for k in range(pk.n_chunks):
        rec = g["c%d_cands" % k]  # pos, lr, var(literal), z, nuc_cov, norm
        cc += [k] * len(rec)
        cp += [int(r[0]) for r in rec]
        ref.append(rec)
    ref = np.concatenate(ref)
    lr, var, z = b.run_candidates(cc, cp)
    assert_track(lr, ref[:, 1], "lr")
    ok = ref[:, 4] > 0
    assert ok.sum() > 0
    assert_track(var[ok], ref[ok, 2], "var")
    assert_track(z[ok], ref[ok, 3], "z")
    b.free()


def test_background_is_formed_on_request(ctx):
    """round 6: the FFT kernel no longer writes T_BACKGROUND (an output only with --write_all, run_nuc.py:22-39); the first request forms
    it from the kernel's two factors per base with the epilogue's own expression.  Requested late (after occ, ins and the candidate
    search), twice, and through the device writer: the reference's values every time, and norm == raw - background bit for bit."""
    g = golden("chunks_basic")
    pk = packed_from_golden(g, True)
    b = ctx.upload(pk)
    b.run_nuc(smooth_sd=10)
    b.run_occ()
    b.run_ins(0, 2000)
    b.run_peaks(min_signal=0, sep=25, boundary=60, order=12)
    bg1 = b.track(L.T_BACKGROUND)
    bg2 = b.track(L.T_BACKGROUND)
    assert np.array_equal(bg1, bg2, equal_nan=True)
    raw, norm = b.track(L.T_RAW), b.track(L.T_NORM)
    assert np.array_equal(norm, raw - bg1, equal_nan=True)
    for k, part in enumerate(b.split(bg1)):
        assert_track(part, g["c%d_bg" % k], "background")
    text, info = b.format_track(L.T_BACKGROUND, ["chrS"] * pk.n_chunks, pk.chunk_start, compress=False)
    assert info["lines"] > 0
    # a second pass over the same batch invalidates and re-forms it
    b.run_nuc(smooth_sd=10)
    assert np.array_equal(b.track(L.T_BACKGROUND), bg1, equal_nan=True)
    b.free()


def test_cython_dropins_edge_cases(ctx):
    g = golden("ins_edge")
    l, n, s, e = g["l"], g["n"], int(g["start"]), int(g["end"])
    for lo, up in ((0, 2000), (0, 251), (2, 251), (100, 300)):
        out = ctx.get_insertions(l, n, s, e, lo, up)
        assert out.dtype == np.float64 and np.array_equal(out, g["ins_%d_%d" % (lo, up)])
        plus, minus = ctx.get_stranded_insertions(l, n, s, e, lo, up)      # getStrandedInsertions, fragments.pyx:71-97
        assert plus.dtype == np.float64 and np.array_equal(plus, g["plus_%d_%d" % (lo, up)])
        assert np.array_equal(minus, g["minus_%d_%d" % (lo, up)])
    ms, me = int(g["mat_start"]), int(g["mat_end"])
    mat = ctx.make_fragment_mat(l, n, ms, me, 0, 251)
    ref = np.zeros_like(mat)
    ref[g["mat_rows"], g["mat_cols"]] = g["mat_vals"]
    assert np.array_equal(mat, ref)
    # the reference's own BAM fixture (tests/test_tracks.py:16-23 of the reference)
    sr = golden("single_read")
    out = ctx.get_insertions(sr["l"], sr["n"], int(sr["start"]), int(sr["end"]), 0, 2000)
    assert np.array_equal(out, sr["ins"]) and out.sum() == 2
    mat = ctx.make_fragment_mat(sr["l"], sr["n"], int(sr["start"]), int(sr["end"]), 0, 100)
    assert np.array_equal(np.array(np.nonzero(mat)).T, sr["mat_nonzero"])
    # empty fragment list
    z = ctx.get_insertions(np.zeros(0, np.int64), np.zeros(0, np.int32), 10, 20)
    assert z.shape == (10,) and not z.any()
    zp, zm = ctx.get_stranded_insertions(np.zeros(0, np.int64), np.zeros(0, np.int32), 10, 20)
    assert zp.shape == zm.shape == (10,) and not zp.any() and not zm.any()


def test_calculate_cov_reference_fixture(ctx):
    """the reference's tests/test_var.py setup (real example bias + example.VMat, r = 35)"""
    g = golden("cov_var_example")
    ref = float(g["var"])
    closed = ctx.calculate_cov(g["p"], g["v"], int(g["r"]))
    literal = ctx.calculate_cov(g["p"], g["v"], int(g["r"]), literal=True)
    assert abs(closed - ref) <= 1e-5 * abs(ref)
    assert abs(literal - ref) <= 1e-5 * abs(ref)
    with pytest.raises(ValueError):
        ctx.calculate_cov(g["p"], g["v"][:-1], 35)
