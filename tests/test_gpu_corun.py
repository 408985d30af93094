"""GPU: natac_run_nuc_occ -- the nuc and occ stages of a batch co-scheduled on the context's two streams (persistent,
half-occupancy launch of the FFT background kernel next to the occupancy stage's kernels, the remaining tiles at full occupancy)
-- gives the bits of natac_run_nuc, natac_run_occ, natac_run_ins called one after the other: every per-base track, the grid
arrays, the candidate arrays, OccPeak arrays, nuc_dist and status words; on repeated steps; when the stop flag comes before the
persistent launch has claimed anything and when the persistent launch claims every tile."""
import os

import numpy as np
import pytest

from helpers import golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution

pytestmark = pytest.mark.gpu

TRACKS = (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH, L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER,
          L.T_OCC_COV, L.T_INS, L.T_OCC_PREFILL)


def _ctx(env):
    from nucleoatac_amd.device import Context
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = Context(0)            # the switches are read when the context is created
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    par = golden("params_example")
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(synth_size_distribution(251))
    nucp, nfrp = synth_occ_distributions(251)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    return ctx


def _everything(b, co, steps=1):
    for _ in range(steps):
        if co:
            b.run_nuc_occ(10, (0, 2000))
        else:
            b.run_nuc(10)
            b.run_occ()
            b.run_ins(0, 2000)
        peaks = b.run_peaks(min_signal=0, sep=25, boundary=60, order=12)
    out = {"t%d" % t: b.track(t) for t in TRACKS}
    for g in (L.G_OCC, L.G_LOWER, L.G_UPPER):
        out["g%d" % g] = b.grid(g)
    for i, a in enumerate(peaks):
        out["pk%d" % i] = a
    for i, a in enumerate(b.run_occ_peaks(min_occ=0.1, sep=120)):
        out["op%d" % i] = a
    out["status"] = b.status()
    return out


def _same(a, b):
    assert a.keys() == b.keys()
    for k in a:
        assert a[k].shape == b[k].shape and a[k].dtype == b[k].dtype, k
        assert np.array_equal(a[k], b[k], equal_nan=(a[k].dtype.kind == "f")), k


def _packed(n_chunks, length, frags, seed, gaps=False):
    counts = np.full(n_chunks, frags, dtype=np.int64)
    if gaps:                                                   # fragment-free chunks: NaN occupancy, the fill, empty candidate lists
        counts[::7] = 0
    return make_synthetic_chunks(n_chunks, length, frags, seed=seed, counts=counts)


@pytest.mark.parametrize("prio", ["0", "3"])
def test_coscheduled_stages_equal_the_stages_one_after_the_other(prio):
    pk = _packed(1500, 2120, 480, seed=11, gaps=True)          # 9,000 background tiles: well above the co-scheduling threshold
    with _ctx({"NATAC_CORUN": "0"}) as ctx:
        b = ctx.upload(pk)
        ref = _everything(b, co=False)
        b.free()
    with _ctx({"NATAC_CORUN": "1", "NATAC_CORUN_PRIO": prio}) as ctx:
        b = ctx.upload(pk)
        got = _everything(b, co=True, steps=3)                 # counters / flag are reset per step
        _same(got, ref)
        again = _everything(b, co=False)                       # and the plain entry points on a two-stream context
        _same(again, ref)
        b.free()


def test_ragged_and_small_batches():
    # a batch below the threshold takes the plain path inside natac_run_nuc_occ; with the threshold at 1 tile the persistent
    # launch sees the stop flag almost at once (tiny occupancy stage) or claims everything (tiny background)
    rng = np.random.default_rng(5)
    for n_chunks, length in ((3, 700), (40, 4000), (300, 1300)):
        pk = _packed(n_chunks, length, int(rng.integers(50, 400)), seed=int(rng.integers(1 << 30)), gaps=n_chunks > 3)
        with _ctx({"NATAC_CORUN": "0"}) as ctx:
            b = ctx.upload(pk)
            ref = _everything(b, co=False)
            b.free()
        for env in ({"NATAC_CORUN": "1"}, {"NATAC_CORUN": "1", "NATAC_CORUN_MIN_TILES": "1"}):
            with _ctx(env) as ctx:
                b = ctx.upload(pk)
                _same(_everything(b, co=True, steps=2), ref)
                b.free()


def test_two_batches_on_one_context_and_release():
    # the streams are joined at the start of each stage: a second batch (or the same one after a release) must not overtake
    pks = [_packed(900, 2120, 300, seed=s) for s in (1, 2)]
    with _ctx({"NATAC_CORUN": "0"}) as ctx:
        refs = []
        for pk in pks:
            b = ctx.upload(pk)
            refs.append(_everything(b, co=False))
            b.free()
    with _ctx({"NATAC_CORUN": "1"}) as ctx:
        bs = [ctx.upload(pk) for pk in pks]
        for b in bs:
            b.run_nuc_occ(10, (0, 2000))
        for b, ref in zip(bs, refs):
            _same(_everything(b, co=True), ref)
            b.release_outputs()
            _same(_everything(b, co=True), ref)
            b.free()
