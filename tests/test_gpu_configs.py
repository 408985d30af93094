"""GPU: BASELINE.json configs[3] (10 kb tiles) and configs[4] (fp64 multinomial_cov path, tolerance sweep)."""
import numpy as np
import pytest

from helpers import assert_track, golden
from nucleoatac_amd import _lib as L
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


def test_config4_10kb_tiles_match_oracle(ctx):
    """hg38-style tiling: 10,000-bp windows (L = 10,120 after slop), ~667 fragments per chunk; a slice of the 300k-chunk
    workload, sample of chunks against the oracle + integer checksums on all of them"""
    from oracle import natac_oracle as O
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    pk = make_synthetic_chunks(3000, 10120, 667, seed=2)
    b = ctx.upload(pk)
    b.run_nuc(10)
    b.run_occ()
    b.run_ins(0, 2000)
    assert not b.status().any()
    nuc_cov, nfr_cov, occ_cov, ins = (b.track(t) for t in (L.T_NUC_COV, L.T_NFR_COV, L.T_OCC_COV, L.T_INS))
    assert np.array_equal(occ_cov, nuc_cov + nfr_cov)
    n = pk.frag_ilen.astype(np.int64)
    l = pk.frag_lpos.astype(np.int64)
    r = l + n - 1
    assert int(ins.sum()) == int(((l >= 0) & (l < 10120)).sum() + ((r >= 0) & (r < 10120)).sum())
    tr = {t: b.track(t) for t in (L.T_NORM, L.T_SMOOTH, L.T_OCC_PREFILL, L.T_OCC_UPPER)}
    for k in (0, 1499, 2999):
        ll, nn = pk.chunk_frags(k)
        ll, nn = ll.astype(np.int64), nn.astype(np.int64)
        a, e = int(pk.out_off[k]), int(pk.out_off[k + 1])
        nt = O.nuc_chunk_tracks(ll, nn, 0, 10120, pk.chunk_bias(k), -pk.bias_left, par["vmat"], 105, 251, sizes)
        oc = O.occ_chunk_tracks(ll, nn, 0, 10120, pk.chunk_bias(k), -pk.bias_left, nucp, nfrp)
        assert_track(tr[L.T_NORM][a:e], nt["norm"], "norm")
        assert_track(tr[L.T_SMOOTH][a:e], nt["smoothed"], "smoothed")
        assert_track(tr[L.T_OCC_PREFILL][a:e], oc["smoothed_vals"], "occ")
        assert_track(tr[L.T_OCC_UPPER][a:e], oc["smoothed_upper"], "occ upper")
    b.free()


def test_config5_multinomial_cov_tolerance_sweep(ctx):
    """BASELINE configs[4] as SURVEY.md section 8(d) cfg 5 defines it: 8 independent samples (seeds) of a configs[2]-like set at 1/8
    size (12,500 chunks each; one per GPU on an 8-GPU node, back to back here).  For every candidate the device finds in a
    200-chunk sample of each set the multinomial variance (nucleoatac/multinomial_cov.pyx:20-31 through SignalDistribution,
    NucleosomeCalling.py:70-86) is computed in three variants on the device -- the .pyx's literal O(N^2) pair sum in fp64, the
    closed form in fp64 (the product path) and the closed form in fp32 -- and compared: max relative error per variant per
    seed against the literal fp64 values, plus the C restatement of the .pyx (the oracle) on a sub-sample.
    Tolerances: closed fp64 vs literal 1e-9; literal vs the oracle's literal 1e-10; fp32 is reported and must be WORSE than
    1e-7 somewhere (it is not good enough for z-scores at the 1e-5 target: the product path stays fp64)."""
    import scale_workers as W
    from test_gpu_properties import _spawn_pool
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    report = []
    oracle_tasks, oracle_meta = [], []
    for seed in range(8):
        pk = make_synthetic_chunks(12500, 2120, 500, seed=1000 + seed)
        b = ctx.upload(pk)
        b.run_nuc(10)
        cc, cp, lr, var, z = b.run_peaks(min_signal=0, sep=25, boundary=60, order=12)
        rng = np.random.default_rng(seed)
        ks = np.sort(rng.choice(pk.n_chunks, size=200, replace=False))
        first, last = np.searchsorted(cc, ks, "left"), np.searchsorted(cc, ks, "right")
        sel = np.concatenate([np.arange(a, e) for a, e in zip(first, last)])
        scc, scp = cc[sel], cp[sel]
        lit = b.run_candidates_cov(scc, scp, "literal")
        clo = b.run_candidates_cov(scc, scp, "closed")
        f32 = b.run_candidates_cov(scc, scp, "fp32")
        ok = lit > 0                                           # candidates with reads (r = 0 gives var = 0 in every variant)
        assert ok.sum() > 3000
        assert np.array_equal(lit[~ok], np.zeros((~ok).sum())) and np.array_equal(clo[~ok], np.zeros((~ok).sum()))
        rel = lambda x: float(np.max(np.abs(x[ok] - lit[ok]) / lit[ok]))
        # the variance natac_run_peaks reported for the same candidates is the closed form
        e_prod = rel(var[sel])
        report.append(dict(seed=seed, candidates=int(len(sel)), with_reads=int(ok.sum()), closed_fp64=rel(clo), fp32=rel(f32),
                           run_peaks_var=e_prod))
        # sub-sample for the oracle: the candidates of 3 chunks per seed
        for k, a, e in list(zip(ks, first, last))[:3]:
            l, n = pk.chunk_frags(int(k))
            oracle_tasks.append((l, n, 2120, pk.chunk_bias(int(k)), pk.bias_left, par["vmat"], 105, 251, sizes, cp[a:e].copy()))
            off = int(np.searchsorted(sel, a))
            oracle_meta.append((seed, lit[off:off + (e - a)].copy(), clo[off:off + (e - a)].copy()))
        b.free()
    with _spawn_pool() as pool:
        ref = pool.map(W.cov_literal_worker, oracle_tasks, chunksize=1)
    worst_lit = worst_clo = 0.0
    n_or = 0
    for (seed, lit, clo), r in zip(oracle_meta, ref):
        m = r[:, 0] > 0
        worst_lit = max(worst_lit, float(np.max(np.abs(lit[m] - r[m, 0]) / r[m, 0])))
        worst_clo = max(worst_clo, float(np.max(np.abs(clo[m] - r[m, 1]) / r[m, 1])))
        n_or += int(m.sum())
    print("configs[4] tolerance sweep, max relative error vs the device's literal fp64 pair sum, per seed:")
    for row in report:
        print("  ", row)
    print("   vs the oracle's C restatement of the .pyx on %d candidates: literal %.3g, closed form %.3g" % (n_or, worst_lit, worst_clo))
    assert n_or > 300
    assert worst_lit < 1e-10 and worst_clo < 1e-9
    for row in report:
        assert row["closed_fp64"] < 1e-9 and row["run_peaks_var"] < 1e-9
        assert 1e-7 < row["fp32"] < 1e-2


def test_calculate_cov_drop_in_variants(ctx):
    """the single-call drop-in (natac_calculate_cov) on 8 random probability windows: literal and closed form against the
    oracle's literal restatement; r is truncated like the .pyx's `int r`"""
    from oracle import natac_oracle as O
    par = golden("params_example")
    v = np.ravel(par["vmat"])
    rng = np.random.default_rng(0)
    worst = {"literal": 0.0, "closed": 0.0}
    for sample in range(8):
        bias = rng.normal(0, 0.8, size=(146, 121))
        p = np.exp(bias) * synth_size_distribution(251)[105:251, None]
        p = (p / p.sum()).ravel()
        r = int(rng.integers(5, 200))
        ref = O.calculate_cov_literal(p, v, r)
        for name, val in (("literal", ctx.calculate_cov(p, v, r, literal=True)), ("closed", ctx.calculate_cov(p, v, r))):
            worst[name] = max(worst[name], abs(val - ref) / abs(ref))
    assert worst["literal"] < 1e-10 and worst["closed"] < 1e-9
    assert ctx.calculate_cov(p, v, 7) == ctx.calculate_cov(p, v, int(7.9))
