"""GPU: size-independent properties at BASELINE.json's full configs[2] size (100k chunks x 2,120 bp, 50 M fragments)
plus ragged / empty / extreme edge cases.  Oracle comparisons are limited to a random sample of chunks."""
import numpy as np
import pytest

from helpers import assert_track, cancel_scale, golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.packing import PackedChunks
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nucleoatac_amd.device import Context
    par = golden("params_example")
    c = Context(0)
    c.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    c.set_sizes(synth_size_distribution(251))
    nucp, nfrp = synth_occ_distributions(251)
    c.set_occ_model(nucp, nfrp, step=5, flank=60)
    yield c
    c.close()


@pytest.fixture(scope="module")
def full(ctx):
    pk = make_synthetic_chunks(100000, 2120, 500, seed=0)
    b = ctx.upload(pk)
    b.run_nuc(10)
    b.run_occ()
    b.run_ins(0, 2000)
    yield pk, b
    b.free()


def test_full_size_integer_checksums(full):
    pk, b = full
    nuc_cov, nfr_cov, occ_cov = b.track(L.T_NUC_COV), b.track(L.T_NFR_COV), b.track(L.T_OCC_COV)
    assert np.array_equal(occ_cov, nuc_cov + nfr_cov)          # same window (121) and size range [0, 251)
    Lc = 2120
    c = pk.frag_lpos.astype(np.int64) + (pk.frag_ilen.astype(np.int64) - 1) // 2
    n = pk.frag_ilen.astype(np.int64)
    ov = np.clip(np.minimum(c + 60, Lc - 1) - np.maximum(c - 60, 0) + 1, 0, None)   # bases whose window holds the centre
    assert int(nuc_cov.sum()) == int(ov[(n >= 105) & (n < 251)].sum())
    assert int(nfr_cov.sum()) == int(ov[(n >= 0) & (n < 105)].sum())
    ins = b.track(L.T_INS)
    l = pk.frag_lpos.astype(np.int64)
    r = l + n - 1
    keep = (n >= 0) & (n < 2000)
    assert int(ins.sum()) == int(((l >= 0) & (l < Lc) & keep).sum() + ((r >= 0) & (r < Lc) & keep).sum())
    assert ins.dtype == np.int32 and ins.min() >= 0


def test_full_size_float_sanity(full):
    pk, b = full
    raw, bg, norm, sm = (b.track(t) for t in (L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH))
    assert np.isfinite(raw).all() and np.isfinite(bg).all() and (bg >= 0).all() and (raw >= 0).all()
    assert np.array_equal(norm, raw - bg)
    assert (sm >= 0).all() and np.isfinite(sm).all()
    occ, lo, hi = (b.track(t) for t in (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER))
    m = ~np.isnan(lo)
    assert m.mean() > 0.99
    assert (occ[m] >= -1e-12).all() and (occ[m] <= 1 + 1e-12).all()
    assert (lo[m] <= occ[m] + 1e-9).all() and (occ[m] <= hi[m] + 1e-9).all()
    g = b.grid(L.G_OCC)
    gm = ~np.isnan(g)
    assert np.all(np.abs(g[gm] * 100 - np.rint(g[gm] * 100)) < 1e-9)   # values of the alpha grid
    assert not b.status().any()


def test_full_size_sample_matches_oracle(full, ctx):
    from oracle import natac_oracle as O
    pk, b = full
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    tr = {t: b.track(t) for t in (L.T_NORM, L.T_SMOOTH, L.T_BACKGROUND, L.T_OCC_PREFILL, L.T_OCC_LOWER, L.T_INS)}
    rng = np.random.default_rng(123)
    for k in rng.integers(0, pk.n_chunks, size=6):
        k = int(k)
        l, n = pk.chunk_frags(k)
        l, n = l.astype(np.int64), n.astype(np.int64)
        a, e = int(pk.out_off[k]), int(pk.out_off[k + 1])
        nt = O.nuc_chunk_tracks(l, n, 0, 2120, pk.chunk_bias(k), -pk.bias_left, par["vmat"], 105, 251, sizes)
        oc = O.occ_chunk_tracks(l, n, 0, 2120, pk.chunk_bias(k), -pk.bias_left, nucp, nfrp)
        assert_track(tr[L.T_BACKGROUND][a:e], nt["bg"], "bg")
        assert_track(tr[L.T_NORM][a:e], nt["norm"], "norm")
        assert_track(tr[L.T_SMOOTH][a:e], nt["smoothed"], "smoothed")
        assert_track(tr[L.T_OCC_PREFILL][a:e], oc["smoothed_vals"], "occ smoothed")
        assert_track(tr[L.T_OCC_LOWER][a:e], oc["smoothed_lower"], "occ lower")
        assert np.array_equal(tr[L.T_INS][a:e], O.get_insertions(l, n, 0, 2120).astype(np.int32))


def test_results_do_not_depend_on_batching(full, ctx):
    """a chunk gives bit-identical tracks alone, in a shard, or in the 100k batch (what sharding over GPUs relies on)"""
    pk, b = full
    sub = pk.subset(4321, 4330)
    sb = ctx.upload(sub)
    sb.run_nuc(10)
    sb.run_occ()
    a, e = int(pk.out_off[4321]), int(pk.out_off[4330])
    for t in (L.T_NORM, L.T_SMOOTH, L.T_BACKGROUND, L.T_OCC, L.T_OCC_UPPER, L.T_NUC_COV):
        assert np.array_equal(sb.track(t), b.track(t)[a:e], equal_nan=True), t
    sb.free()
    again = b.track(L.T_NORM).copy()
    b.run_nuc(10)
    assert np.array_equal(again, b.track(L.T_NORM))           # deterministic


def test_zero_bias_equals_no_bias(ctx):
    pk = make_synthetic_chunks(50, 900, 200, seed=7)
    zero = PackedChunks(pk.chunk_start, pk.chunk_len, pk.frag_off, pk.frag_lpos, pk.frag_ilen, pk.bias_off,
                        np.zeros_like(pk.bias_log))
    none = PackedChunks(pk.chunk_start, pk.chunk_len, pk.frag_off, pk.frag_lpos, pk.frag_ilen, None, None)
    outs = []
    for p in (zero, none):
        b = ctx.upload(p)
        b.run_nuc(10)
        b.run_occ()
        outs.append([b.track(t) for t in (L.T_BACKGROUND, L.T_NORM, L.T_OCC, L.T_OCC_LOWER)])
        b.free()
    for x, y in zip(*outs):
        assert np.array_equal(x, y, equal_nan=True)


def test_duplicated_fragments_scale_signal(ctx):
    pk = make_synthetic_chunks(20, 800, 150, seed=8)
    rep = lambda a: np.repeat(a, 2)
    dbl = PackedChunks(pk.chunk_start, pk.chunk_len, pk.frag_off * 2, rep(pk.frag_lpos), rep(pk.frag_ilen), pk.bias_off,
                       pk.bias_log)
    res = []
    for p in (pk, dbl):
        b = ctx.upload(p)
        b.run_nuc(10)
        b.run_ins(0, 2000)
        res.append([b.track(t) for t in (L.T_NUC_COV, L.T_RAW, L.T_BACKGROUND, L.T_INS)])
        b.free()
    assert np.array_equal(res[1][0], 2 * res[0][0]) and np.array_equal(res[1][3], 2 * res[0][3])
    np.testing.assert_allclose(res[1][1], 2 * res[0][1], rtol=1e-13)
    np.testing.assert_allclose(res[1][2], 2 * res[0][2], rtol=1e-13)


def test_ragged_and_empty_chunks_match_oracle(ctx):
    """ragged lengths (incl. the minimum 121 and L % 5 != 0), a chunk without fragments, fragments outside every window,
    huge insert sizes, a crowded chunk (> 1024 fragments per occupancy tile: un-staged path)"""
    from oracle import natac_oracle as O
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    rng = np.random.default_rng(5)
    lens = [121, 122, 333, 1204, 700, 640]
    fr = []
    for i, Lc in enumerate(lens):
        if i == 2:
            l, n = np.zeros(0, np.int64), np.zeros(0, np.int64)           # no fragments at all
        elif i == 5:
            n = rng.integers(20, 300, size=12000)                          # crowded: ~15 fragments per bp
            l = rng.integers(-150, Lc + 100, size=12000)
        else:
            n = np.concatenate((rng.integers(1, 400, size=60 + i * 40), [1, 2, 0, 1999, 2500, 250, 251]))
            l = rng.integers(-300, Lc + 200, size=len(n))
        c = l + (n - 1) // 2
        o = np.argsort(c, kind="stable")
        fr.append((l[o], n[o]))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    nb = [Lc + 493 for Lc in lens]
    bias = rng.normal(0, 0.7, size=sum(nb))
    pk = PackedChunks(np.arange(len(lens)) * 5000, lens, off, np.concatenate([x[0] for x in fr]),
                      np.concatenate([x[1] for x in fr]), np.concatenate(([0], np.cumsum(nb))), bias)
    b = ctx.upload(pk)
    b.run_nuc(10)
    b.run_occ()
    b.run_ins(0, 2000)
    assert not b.status().any()
    tr = {t: b.split(b.track(t)) for t in (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH,
                                           L.T_OCC_PREFILL, L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV, L.T_INS)}
    for k, Lc in enumerate(lens):
        l, n = fr[k]
        nt = O.nuc_chunk_tracks(l, n, 0, Lc, pk.chunk_bias(k), -246, par["vmat"], 105, 251, sizes)
        oc = O.occ_chunk_tracks(l, n, 0, Lc, pk.chunk_bias(k), -246, nucp, nfrp)
        assert_track(tr[L.T_NUC_COV][k], nt["nuc_cov"], "nuc_cov", exact=True)
        assert_track(tr[L.T_NFR_COV][k], nt["nfr_cov"], "nfr_cov", exact=True)
        assert_track(tr[L.T_RAW][k], nt["raw"], "raw")
        assert_track(tr[L.T_BACKGROUND][k], nt["bg"], "bg")
        sc = cancel_scale(nt["raw"], nt["bg"])
        assert_track(tr[L.T_NORM][k], nt["norm"], "norm", scale=sc)
        assert_track(tr[L.T_SMOOTH][k], nt["smoothed"], "smoothed", scale=sc)
        assert_track(tr[L.T_OCC_PREFILL][k], oc["smoothed_vals"], "occ")
        assert_track(tr[L.T_OCC_LOWER][k], oc["smoothed_lower"], "occ lower")
        assert_track(tr[L.T_OCC_UPPER][k], oc["smoothed_upper"], "occ upper")
        assert_track(tr[L.T_OCC_COV][k], oc["cov"], "occ cov", exact=True)
        assert np.array_equal(tr[L.T_INS][k], O.get_insertions(l, n, 0, Lc).astype(np.int32))
        filled = oc["smoothed_vals"].copy()
        O.call_peaks(filled)
        assert_track(tr[L.T_OCC][k], filled, "occ post-fill")
    assert np.isnan(tr[L.T_OCC][2]).all() and not tr[L.T_NUC_COV][2].any()       # the empty chunk stays NaN / 0
    b.free()


def test_argument_errors(ctx):
    pk = make_synthetic_chunks(3, 100, 10, seed=1)            # shorter than the 121-bp windows
    b = ctx.upload(pk)
    with pytest.raises(L.NatacError):
        b.run_occ()                                            # 100 < 121
    with pytest.raises(L.NatacError):
        b.run_nuc(20)                                          # 100 < 6*20+1
    with pytest.raises(L.NatacError):
        b.track(L.T_NORM)                                      # stage has not run
    b.run_ins()
    with pytest.raises(L.NatacError):
        b.run_candidates([0], [5])                             # needs run_nuc first
    b.free()
    ok = make_synthetic_chunks(2, 400, 50, seed=1)
    b = ctx.upload(ok)
    b.run_nuc(10)
    with pytest.raises(L.NatacError):
        b.run_candidates([5], [10])                            # chunk index out of range
    assert b.run_candidates([], [])[0].shape == (0,)
    b.free()
    short = PackedChunks(ok.chunk_start, ok.chunk_len, ok.frag_off, ok.frag_lpos, ok.frag_ilen,
                         np.arange(3) * (400 + 200), np.zeros(2 * 600), bias_left=100, bias_right=100)
    b = ctx.upload(short)
    with pytest.raises(L.NatacError):
        b.run_nuc(10)                                          # bias halo too small for the 185-bp reach
    b.free()


def test_zero_probability_bins_reproduce_reference_failure():
    """a size bin with probability 0 in BOTH distributions makes every log-likelihood -inf (0*log 0 = NaN -> -inf,
    Occupancy.py:112-114): the reference dies with ValueError at :118; the library flags the chunk and writes NaN.
    A zero in only ONE distribution just forces alpha = 0 / 1 to -inf."""
    from nucleoatac_amd.device import Context
    from oracle import natac_oracle as O
    pk = make_synthetic_chunks(2, 400, 120, seed=4)
    nucp, nfrp = synth_occ_distributions(251)
    l0, n0 = pk.chunk_frags(0)
    with Context(0) as c:
        both = (nucp.copy(), nfrp.copy())
        both[0][7] = 0.0
        both[1][7] = 0.0
        c.set_occ_model(both[0] / both[0].sum(), both[1] / both[1].sum(), step=5, flank=60)
        b = c.upload(pk)
        b.run_occ()
        assert (b.status() & 1).all() and np.isnan(b.grid(L.G_OCC)).all()
        with pytest.raises(ValueError):
            O.occ_chunk_tracks(l0.astype(np.int64), n0.astype(np.int64), 0, 400, pk.chunk_bias(0), -246,
                               both[0] / both[0].sum(), both[1] / both[1].sum())
        b.free()
        one = nucp.copy()
        one[7] = 0.0
        one /= one.sum()
        c.set_occ_model(one, nfrp, step=5, flank=60)
        b = c.upload(pk)
        b.run_occ()
        assert not b.status().any()
        oc = O.occ_chunk_tracks(l0.astype(np.int64), n0.astype(np.int64), 0, 400, pk.chunk_bias(0), -246, one, nfrp)
        g = b.grid(L.G_UPPER)[:len(range(2, 400, 5))]
        assert np.array_equal(np.repeat(g, 5)[:400], oc["occ_upper"], equal_nan=True) and np.nanmax(g) < 1.0
        b.free()


def test_background_fft_fallback_tiles_match_oracle(ctx):
    """bias windows that the FFT background kernel must not transform (NaN, -inf, a huge dynamic range) are evaluated by
    direct summation inside the kernel: NaNs stay confined to the bases whose window touches them, exactly where the
    reference's dense correlation has them, and every other base agrees with the oracle."""
    from oracle import natac_oracle as O
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    rng = np.random.default_rng(11)
    lens = [1500, 1500, 1500, 1500]
    fr = []
    for Lc in lens:
        n = rng.integers(30, 300, size=700)
        l = rng.integers(-150, Lc + 100, size=700)
        o = np.argsort(l + (n - 1) // 2, kind="stable")
        fr.append((l[o], n[o]))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    nb = [Lc + 493 for Lc in lens]
    boff = np.concatenate(([0], np.cumsum(nb)))
    bias = rng.normal(0, 0.7, size=sum(nb))
    bias[boff[0] + 246 + 700] = np.nan                     # chunk 0: one NaN in the middle
    bias[boff[1] + 246 + 300] = -np.inf                    # chunk 1: exp(-inf) = 0
    bias[boff[2] + 246 + 900:boff[2] + 246 + 1000] += 25.0   # chunk 2: dynamic range e^25 inside one tile
    bias[boff[3]:boff[4]] = rng.uniform(-5.0, 4.9, size=nb[3])   # chunk 3: e^9.9 = 2e4, the widest range still transformed
    pk = PackedChunks(np.arange(len(lens)) * 5000, lens, off, np.concatenate([x[0] for x in fr]),
                      np.concatenate([x[1] for x in fr]), boff, bias)
    b = ctx.upload(pk)
    b.run_nuc(10)
    bg = b.split(b.track(L.T_BACKGROUND))
    nm = b.split(b.track(L.T_NORM))
    for k, Lc in enumerate(lens):
        l, n = fr[k]
        with np.errstate(all="ignore"):
            nt = O.nuc_chunk_tracks(l, n, 0, Lc, pk.chunk_bias(k), -246, par["vmat"], 105, 251, sizes)
        assert np.array_equal(np.isnan(bg[k]), np.isnan(nt["bg"])), k
        assert_track(bg[k], nt["bg"], "bg %d" % k)
        assert_track(nm[k], nt["norm"], "norm %d" % k, scale=cancel_scale(nt["raw"], nt["bg"]))
    assert np.isnan(bg[0]).sum() > 0 and np.isnan(bg[0]).sum() < 600 and not np.isnan(bg[3]).any()
    b.free()


def test_background_fft_equals_direct_kernel(ctx):
    """NATAC_BG_DIRECT=1 (direct summation, the pre-FFT kernel) and the FFT kernel agree to ~1e-13 on a ragged batch"""
    import os
    import subprocess
    import sys
    import tempfile
    code = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.device import Context
from nucleoatac_amd.synth import make_synthetic_chunks, synth_size_distribution
par = golden("params_example")
c = Context(0); c.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"])); c.set_sizes(synth_size_distribution(251))
pk = make_synthetic_chunks(300, 2120, 500, seed=3)
b = c.upload(pk); b.run_nuc(10)
np.save(sys.argv[1], np.stack([b.track(L.T_BACKGROUND), b.track(L.T_NORM), b.track(L.T_SMOOTH)]))
b.free(); c.close()
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    out = []
    with tempfile.TemporaryDirectory() as td:
        for mode in ("0", "1"):
            env = dict(os.environ, NATAC_BG_DIRECT=mode)
            path = os.path.join(td, "m%s.npy" % mode)
            subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
            out.append(np.load(path))
    for t in range(3):
        np.testing.assert_allclose(out[0][t], out[1][t], rtol=1e-10, atol=1e-12)
    assert not np.array_equal(out[0][0], out[1][0])          # really two different kernels


def _spawn_pool():
    """oracle workers in spawned processes (this process already holds a HIP context: no fork)"""
    import multiprocessing as mp
    import os
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(q) // int(p)))
    except (OSError, ValueError):
        pass
    return mp.get_context("spawn").Pool(max(1, min(n, 32) - 1))


def test_full_size_occupancy_grid_exact_on_2000_chunks(full):
    """the MLE decision at scale: the raw occupancy grid (alpha indices: vals / lower_bound / upper_bound, Occupancy.py:104-146)
    of 2,000 random chunks of the configs[2] batch (848,000 grid points) against the oracle's literal log-likelihood
    arithmetic, EXACTLY.  The kernel decides in the product domain (frexp-renormalised likelihood products), so this
    measures how often a near-tie in argmax / the likelihood-ratio test flips an index: it must be zero."""
    import scale_workers as W
    pk, b = full
    nucp, nfrp = synth_occ_distributions(251)
    rng = np.random.default_rng(2024)
    ks = np.sort(rng.choice(pk.n_chunks, size=2000, replace=False))
    tasks = []
    for k in ks:
        l, n = pk.chunk_frags(int(k))
        tasks.append((l, n, int(pk.chunk_len[k]), pk.chunk_bias(int(k)), pk.bias_left, nucp, nfrp))
    with _spawn_pool() as pool:
        ref = pool.map(W.occ_grid_worker, tasks, chunksize=8)
    grids = [b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)]
    nk = len(range(2, 2120, 5))
    flips = [0, 0, 0]
    npts = 0
    for k, r in zip(ks, ref):
        a = int(k) * nk
        for w in range(3):
            got, want = grids[w][a:a + nk], r[w]
            assert got.shape == want.shape
            assert np.array_equal(np.isnan(got), np.isnan(want)), (int(k), w)
            m = ~np.isnan(want)
            flips[w] += int(np.sum(got[m] != want[m]))
        npts += nk
    print("occupancy grid at scale: %d grid points of %d chunks; flipped alpha indices occ/lower/upper = %s" % (npts, len(ks), flips))
    assert flips == [0, 0, 0]


def test_full_size_candidates_match_oracle_on_200_chunks(full):
    """candidate statistics at scale: every candidate the device finds in 200 random chunks of the configs[2] batch gets the
    oracle's lr / var / z (rtol 1e-5), and the oracle's call_peaks candidates above the FFT noise floor are all found"""
    import scale_workers as W
    pk, b = full
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    cc, cp, lr, var, z = b.run_peaks(min_signal=0, sep=25, boundary=60, order=12)
    assert len(cc) > 1500000
    rng = np.random.default_rng(77)
    ks = np.sort(rng.choice(pk.n_chunks, size=200, replace=False))
    first = np.searchsorted(cc, ks, "left")
    last = np.searchsorted(cc, ks, "right")
    tasks = []
    for k, a, e in zip(ks, first, last):
        l, n = pk.chunk_frags(int(k))
        tasks.append((l, n, int(pk.chunk_len[k]), pk.chunk_bias(int(k)), pk.bias_left, par["vmat"], 105, 251, sizes, cp[a:e].copy()))
    with _spawn_pool() as pool:
        res = pool.map(W.cand_worker, tasks, chunksize=2)
    ncand = nz = 0
    for (ref_c, ref_sig, st), a, e in zip(res, first, last):
        mine = set(int(x) for x in cp[a:e])
        assert set(int(x) for x in ref_c[ref_sig > 1e-9]) <= mine
        assert_track(lr[a:e], st[:, 0], "lr")
        ok = st[:, 3] > 0
        assert_track(var[a:e][ok], st[ok, 1], "var")
        assert_track(z[a:e][ok], st[ok, 2], "z")
        ncand += e - a
        nz += int(ok.sum())
    print("candidates at scale: %d candidates of 200 chunks compared (lr), %d with reads (var, z)" % (ncand, nz))
    assert ncand > 3000


def test_fast_occupancy_defers_ill_conditioned_tiles_to_the_general_kernel(ctx):
    """exp(bias) values outside 2^+-190 (or non-finite) make natac_occ_gsum poison its blocks; natac_occ_decide then hands the
    tiles that touch them to the general kernel natac_occ_mle through the device-side list.  The result must equal a run
    that uses the general kernel for every tile (NATAC_OCC_GENERAL=1) bit for bit -- grid values, smoothed tracks, status
    words.  Against the oracle: exact for the tiny values (and everywhere outside the touched windows); a huge value
    (e^135 next to e^0) is beyond the general kernel's sliding window sums -- those grid points come back NaN with the
    chunk's status bit set (flagged, never silent); a NaN bias likewise (the reference raises ValueError there)."""
    import os
    from nucleoatac_amd.device import Context
    from oracle import natac_oracle as O
    pk = make_synthetic_chunks(6, 2120, 500, seed=21)
    bias = pk.bias_log.copy()
    per = 2120 + pk.bias_left + pk.bias_right
    bias[2 * per + 900:2 * per + 905] = -400.0            # exp() = 1e-174 < 2^-190: tiny but finite
    bias[3 * per + 1500] = 135.0                           # exp() = 4e58 > 2^190: huge
    bias[4 * per + 700] = np.nan                           # the reference's likelihood is undefined around this base
    bias[5 * per + 1200] = 100.0                           # exp() = 2.7e43: extreme but inside the fast path's range
    bad = PackedChunks(pk.chunk_start, pk.chunk_len, pk.frag_off, pk.frag_lpos, pk.frag_ilen, pk.bias_off, bias)
    nucp, nfrp = synth_occ_distributions(251)

    def run(c):
        b = c.upload(bad)
        b.run_occ()
        out = ([b.grid(w) for w in (L.G_OCC, L.G_LOWER, L.G_UPPER)], [b.track(t) for t in (L.T_OCC_PREFILL, L.T_OCC_LOWER, L.T_OCC_UPPER)],
               b.status())
        b.free()
        return out

    fast = run(ctx)
    os.environ["NATAC_OCC_GENERAL"] = "1"
    try:
        par = golden("params_example")
        g = Context(0)
        g.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
        g.set_sizes(synth_size_distribution(251))
        g.set_occ_model(nucp, nfrp, step=5, flank=60)
        general = run(g)
        g.close()
    finally:
        del os.environ["NATAC_OCC_GENERAL"]
    nk = len(range(2, 2120, 5))
    for k in (0, 1, 2, 3, 4):                              # chunk 5 (e^100) stays on the fast path: compared with the oracle below
        for a, b in zip(fast[0], general[0]):
            assert np.array_equal(a[k * nk:(k + 1) * nk], b[k * nk:(k + 1) * nk], equal_nan=True), k
    assert np.array_equal(fast[2][:5], general[2][:5]) and fast[2][4] & 1 and fast[2][3] & 1 and not fast[2][[0, 1, 2, 5]].any()
    for k in (0, 2, 3, 5):                                 # chunk 4 makes the reference raise ValueError (Occupancy.py:118)
        l, n = bad.chunk_frags(k)
        oc = O.occ_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, 2120, bad.chunk_bias(k), -bad.bias_left, nucp, nfrp)
        for w, key in enumerate(("occ", "occ_lower", "occ_upper")):
            got, ref = fast[0][w][k * nk:(k + 1) * nk], oc[key][2::5]
            if k == 3:      # flagged NaNs inside the windows around the huge value, exact elsewhere
                far = np.abs(2 + 5 * np.arange(nk) - (1500 - bad.bias_left)) > 400
                assert np.array_equal(got[far], ref[far], equal_nan=True) and np.isnan(got[~far]).any(), key
            else:
                assert np.array_equal(got, ref, equal_nan=True), (k, key)
