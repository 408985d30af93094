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
    """calculateCov variants on 8 independent synthetic samples (one per GPU in the 8-GPU config): literal O(N^2) fp64 on
    the GPU, closed form fp64 on the GPU, and an fp32 closed form for reference; errors are relative to the oracle's literal
    restatement of the .pyx.  Tolerances: literal 1e-10, closed 1e-9 (north-star float tolerance is 1e-5)."""
    from oracle import natac_oracle as O
    par = golden("params_example")
    v = np.ravel(par["vmat"])
    rng = np.random.default_rng(0)
    worst = {"literal": 0.0, "closed": 0.0, "fp32": 0.0}
    for sample in range(8):
        bias = rng.normal(0, 0.8, size=(146, 121))
        p = np.exp(bias) * synth_size_distribution(251)[105:251, None]
        p = (p / p.sum()).ravel()
        r = int(rng.integers(5, 200))
        ref = O.calculate_cov_literal(p, v, r)
        lit = ctx.calculate_cov(p, v, r, literal=True)
        clo = ctx.calculate_cov(p, v, r)
        p32, v32 = p.astype(np.float32), v.astype(np.float32)
        f32 = float(np.float32(r) * (np.sum(p32 * v32 * v32) - np.sum(p32 * v32) ** 2))
        for name, val in (("literal", lit), ("closed", clo), ("fp32", f32)):
            worst[name] = max(worst[name], abs(val - ref) / abs(ref))
    print("max relative error vs the .pyx restatement:", worst)
    assert worst["literal"] < 1e-10 and worst["closed"] < 1e-9
    assert worst["fp32"] < 1e-2          # fp32 is NOT good enough for the 1e-5 target: the product path stays fp64
    # r is truncated like the .pyx's `int r`
    assert ctx.calculate_cov(p, v, 7) == ctx.calculate_cov(p, v, int(7.9))
