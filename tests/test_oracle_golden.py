"""CPU: the oracle (oracle/natac_oracle.py) against the golden vectors that the REFERENCE produced
(tests/golden/*.npz, generator tests/golden/make_golden.py).  This is the oracle's pin."""
import numpy as np
import pytest

from helpers import assert_track, golden
from oracle import natac_oracle as O

CASES = [("chunks_basic", True), ("chunks_gaps", True), ("chunks_nobias", False)]


@pytest.fixture(scope="module")
def par():
    return golden("params_example")


@pytest.mark.parametrize("case,with_bias", CASES)
def test_nuc_tracks(case, with_bias, par):
    g = golden(case)
    for k in range(int(g["n_chunks"])):
        s, e = int(g["chunk_start"][k]), int(g["chunk_end"][k])
        bias = g["c%d_bias_log" % k] if with_bias else None
        nt = O.nuc_chunk_tracks(g["c%d_l" % k], g["c%d_n" % k], s, e, bias, s - 246, par["vmat"], int(par["vlower"]),
                                int(par["vupper"]), par["sizes"], smooth_sd=10)
        assert_track(nt["nuc_cov"], g["c%d_nuc_cov" % k], "nuc_cov", exact=True)
        assert_track(nt["nfr_cov"], g["c%d_nfr_cov" % k], "nfr_cov", exact=True)
        assert_track(nt["raw"], g["c%d_raw" % k], "raw", rtol=1e-9, atol=1e-12)
        assert_track(nt["bg"], g["c%d_bg" % k], "bg", rtol=1e-9, atol=1e-12)
        assert_track(nt["norm"], g["c%d_norm" % k], "norm", rtol=1e-9, atol=1e-11)
        assert_track(nt["smoothed"], g["c%d_smoothed" % k], "smoothed", rtol=1e-9, atol=1e-11)
        ins, half = O.get_ins_from_mat(nt["mat"], 0, 251)
        assert_track(ins, g["c%d_getins" % k], "getIns", exact=True)
        assert_track(O.get_insertions(g["c%d_l" % k], g["c%d_n" % k], s, e, 0, 2000), g["c%d_ins2000" % k], "ins", exact=True)
        assert_track(O.get_insertions(g["c%d_l" % k], g["c%d_n" % k], s, e, 0, 251), g["c%d_ins251" % k], "ins251", exact=True)
        # candidates: call_peaks + LR + z
        rec = g["c%d_cands" % k]
        comb = nt["norm"] + nt["smoothed"]
        cands = O.call_peaks(comb.copy(), min_signal=0, sep=25, boundary=60, order=12)
        cands = np.array([c for c in cands if comb[c] > 1e-9])
        assert np.array_equal(cands, rec[:, 0].astype(np.int64))
        for row in rec[: 6]:
            pos = int(row[0]) + s
            lr = O.get_lr(nt["mat"], nt["mat_start"], nt["bmat"], nt["b0"], nt["b_start"], par["vmat"], int(par["vlower"]),
                          int(par["vupper"]), pos)
            assert abs(lr - row[1]) <= 1e-8 * max(1.0, abs(row[1]))
            if row[4] > 0:
                pr = O.signal_distribution_probs(nt["bmat"], nt["b_start"], int(par["vlower"]), int(par["vupper"]), 60, pos)
                z, var = O.z_score(row[5], row[4], pr, par["vmat"])
                assert abs(var - row[2]) <= 1e-7 * abs(row[2])
                assert abs(z - row[3]) <= 1e-7 * abs(row[3])


@pytest.mark.parametrize("case,with_bias", CASES)
def test_occ_tracks(case, with_bias, par):
    g = golden(case)
    for k in range(int(g["n_chunks"])):
        s, e = int(g["chunk_start"][k]), int(g["chunk_end"][k])
        bias = g["c%d_bias_log" % k] if with_bias else None
        oc = O.occ_chunk_tracks(g["c%d_l" % k], g["c%d_n" % k], s, e, bias, s - 246, par["nuc_probs"], par["nfr_probs"])
        assert_track(oc["occ"], g["c%d_occ" % k], "occ", exact=True)
        assert_track(oc["occ_lower"], g["c%d_occ_lower" % k], "occ_lower", exact=True)
        assert_track(oc["occ_upper"], g["c%d_occ_upper" % k], "occ_upper", exact=True)
        assert_track(oc["smoothed_vals"], g["c%d_occ_smoothed_prefill" % k], "smoothed", rtol=1e-12)
        assert_track(oc["smoothed_lower"], g["c%d_occ_smoothed_lower" % k], "smoothed_lower", rtol=1e-12)
        assert_track(oc["smoothed_upper"], g["c%d_occ_smoothed_upper" % k], "smoothed_upper", rtol=1e-12)
        assert_track(oc["cov"], g["c%d_occ_cov" % k], "cov", exact=True)
        filled = oc["smoothed_vals"].copy()
        pk = O.call_peaks(filled, sep=120, min_signal=0.1)
        assert_track(filled, g["c%d_occ_smoothed" % k], "smoothed post-fill", rtol=1e-12)
        keep = np.array([p for p in pk if oc["smoothed_lower"][p] > 0.1 and oc["cov"][p] > 0], dtype=np.int64)
        assert np.array_equal(keep, g["c%d_occ_peaks" % k])


def test_gaps_case_has_nan_fill():
    """chunks_gaps exercises the call_peaks in-place NaN fill (reference quirk)"""
    g = golden("chunks_gaps")
    assert np.isnan(g["c0_occ_smoothed_prefill"]).sum() > 0
    assert np.isnan(g["c0_occ_smoothed"]).sum() == 0
    assert np.isnan(g["c0_occ_smoothed_lower"]).sum() > 0


def test_bias_from_sequence(par):
    g = golden("chunks_basic")
    nucs = [str(x) for x in par["pwm_nucleotides"]]
    for k in range(int(g["n_chunks"])):
        seq = bytes(g["c%d_seq" % k]).decode()
        assert "N" in seq or k != 1
        b = O.compute_bias_pwm(seq, par["pwm_mat"], nucs)
        assert_track(b, g["c%d_bias_log" % k], "bias", rtol=1e-12, atol=1e-12)


def test_calculate_cov_reference_fixture():
    """the reference's own tests/test_var.py setup"""
    g = golden("cov_var_example")
    lit = O.calculate_cov_literal(g["p"], g["v"], int(g["r"]))
    clo = O.calculate_cov_closed(g["p"], g["v"], int(g["r"]))
    assert abs(lit - float(g["var"])) <= 1e-12 * abs(float(g["var"]))
    assert abs(clo - float(g["var"])) <= 1e-9 * abs(float(g["var"]))
    with pytest.raises(ValueError):
        O.calculate_cov_literal(g["p"], g["v"][:-1], 35)
    bm = O.make_bias_mat(g["bias_track"], int(g["bias_track_start"]), int(g["biasmat_start"]), int(g["biasmat_end"]),
                         int(g["biasmat_lower"]), int(g["biasmat_upper"]))
    assert_track(bm[g["biasmat_sample_rows"]], g["biasmat_samples"], "BiasMat2D rows", rtol=1e-13)


def test_toy_occupancy():
    """the reference's tests/test_occupancy.py:15-22 cases + 40 random draws"""
    g = golden("toy_occupancy")
    for ins, bias, ref in zip(g["ins"], g["bias"], g["result"]):
        got = O.calculate_occupancy(ins, bias, g["nuc_probs"], g["nfr_probs"], g["alphas"], float(g["cutoff"]))
        assert tuple(got) == tuple(ref)
    assert g["result"][0][0] == 0 and g["result"][1][0] == 0.5


def test_insertion_edge_cases():
    g = golden("ins_edge")
    l, n, s, e = g["l"], g["n"], int(g["start"]), int(g["end"])
    for lo, up in ((0, 2000), (0, 251), (2, 251), (100, 300)):
        assert np.array_equal(O.get_insertions(l, n, s, e, lo, up), g["ins_%d_%d" % (lo, up)])
        plus, minus = O.get_stranded_insertions(l, n, s, e, lo, up)
        assert np.array_equal(plus, g["plus_%d_%d" % (lo, up)]) and np.array_equal(minus, g["minus_%d_%d" % (lo, up)])
    mat = O.make_fragment_mat(l, n, int(g["mat_start"]), int(g["mat_end"]), 0, 251)
    ref = np.zeros_like(mat)
    ref[g["mat_rows"], g["mat_cols"]] = g["mat_vals"]
    assert np.array_equal(mat, ref)
    ins, half = O.get_ins_from_mat(mat, 0, 251)
    assert np.array_equal(ins, g["getins"]) and int(g["mat_start"]) + half == int(g["getins_start"])
    sr = golden("single_read")
    assert np.array_equal(O.get_insertions(sr["l"], sr["n"], int(sr["start"]), int(sr["end"])), sr["ins"])


def test_smooth_matches_numpy_semantics():
    rng = np.random.default_rng(0)
    x = rng.normal(size=300)
    x[40:70] = np.nan
    y = O.smooth(x, 61, window="gaussian", sd=10, mode="same", norm=True)
    assert y.shape == x.shape and np.isnan(y).sum() == 0
    x[:] = np.nan
    assert np.isnan(O.smooth(x, 61, window="gaussian", sd=10, mode="same")).all()
    assert O.smooth(np.ones(10), 4, mode="valid", norm=False).shape == (6,)  # even window is bumped to 5


def test_fragment_size_histogram_matches_reference():
    """a3: getFragmentSizesFromChunkList / getAllFragmentSizes / calculateSizes (pyatac/fragments.pyx:101-145,
    fragmentsizes.py:22-27) on overlapping, nested, adjacent, empty, chromosome-start and unsorted chunk lists"""
    g = golden("sizes_hist")
    n_checked = 0
    for lo, up, atac in ((0, 251, 1), (30, 251, 1), (0, 2000, 1), (105, 251, 0)):
        tag = "%d_%d_%d" % (lo, up, atac)
        l, n = (g["l"], g["n"]) if atac else (g["l"] - 4, g["n"] + 8)
        assert np.array_equal(O.fragment_sizes_from_chunks(l, n, [-(1 << 40)], [1 << 40], lo, up), g["all_" + tag])
        for name in ("disjoint", "overlap", "adjacent", "empty", "chromstart", "unsorted"):
            iv = g[name + "_chunks"]
            got = O.fragment_sizes_from_chunks(l, n, iv[:, 0], iv[:, 1], lo, up)
            assert np.array_equal(got, g["%s_%s" % (name, tag)]), (name, tag)
            assert np.array_equal(O.normalise_sizes(got), g["%s_%s_norm" % (name, tag)])
            n_checked += 1
    assert n_checked == 24 and g["overlap_0_251_1"].sum() > 0 and g["empty_0_251_1"].sum() == 0
