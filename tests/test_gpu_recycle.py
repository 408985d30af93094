"""GPU: BASELINE.json configs[3] on ONE GPU takes the output-recycling path -- a shard of sub-batches whose inputs stay resident
while every sub-batch's outputs are released after its stages (natac_batch_release_outputs, nucleoatac_amd/executor.py::
ResidentShard).  A released batch must behave exactly like a fresh one: second pass == first pass == no-recycle pass, bit for bit,
and the values themselves are checked against the oracle."""
import numpy as np
import pytest

from helpers import assert_track, golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.executor import ResidentShard, Stages
from nucleoatac_amd.synth import fragment_counts, make_synthetic_chunks, synth_occ_distributions, synth_size_distribution

pytestmark = pytest.mark.gpu

TL = 10120        # configs[3] tile length after the +-60 slop
TRACKS = {"nuc_cov": L.T_NUC_COV, "nfr_cov": L.T_NFR_COV, "raw": L.T_RAW, "bg": L.T_BACKGROUND, "norm": L.T_NORM,
          "smoothed": L.T_SMOOTH, "occ": L.T_OCC_PREFILL, "occ_filled": L.T_OCC, "occ_lower": L.T_OCC_LOWER,
          "occ_upper": L.T_OCC_UPPER, "occ_cov": L.T_OCC_COV, "ins": L.T_INS}


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


def _subs(n_sub=4, per=30):
    """cfg4-shaped sub-batches: 10,120-bp tiles, Poisson(667) fragments, generated in counter-seeded blocks like bench.py"""
    counts = fragment_counts(n_sub * per, 667, seed=2)
    return [make_synthetic_chunks(per, TL, 667, seed=[2, i], counts=counts[i * per:(i + 1) * per], first_chunk=i * per)
            for i in range(n_sub)]


def _collect(store):
    def consume(i, b, n):
        out = {name: b.track(t) for name, t in TRACKS.items()}
        for w, g in (("g_occ", L.G_OCC), ("g_lower", L.G_LOWER), ("g_upper", L.G_UPPER)):
            out[w] = b.grid(g)
        out["peaks"] = b.download_peaks(n)
        out["occ_peaks"] = b.run_occ_peaks(min_occ=0.1, sep=120)
        out["status"] = b.status()
        store[i] = out
    return consume


def _same(a, b, what):
    if isinstance(a, tuple):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            _same(x, y, what)
    else:
        assert a.dtype == b.dtype and a.shape == b.shape, what
        assert np.array_equal(a, b, equal_nan=a.dtype.kind == "f"), what


def test_released_batches_recompute_bit_identically_and_match_the_oracle(ctx):
    import scale_workers as W
    from test_gpu_properties import _spawn_pool
    subs = _subs()
    stages = Stages(nuc_sd=10, occ=True, ins=(0, 2000), peaks=dict(min_signal=0, sep=25, boundary=60, order=12))
    shard = ResidentShard(ctx, subs, recycle=True)
    assert shard.recycle
    p1, p2, p3 = {}, {}, {}
    n1 = shard.step(stages, consume=_collect(p1))
    # after the release nothing of the outputs is left on the device: a download is a state error, not stale data
    for b in shard.batches:
        with pytest.raises(L.NatacError):
            b.track(L.T_NORM)
        with pytest.raises(L.NatacError):
            b.grid(L.G_OCC)
    n2 = shard.step(stages, consume=_collect(p2))         # second pass through released batches
    shard.close()
    fresh = ResidentShard(ctx, subs, recycle=False)        # and the same stages without any release
    n3 = fresh.step(stages, consume=_collect(p3))
    fresh.close()
    assert n1 == n2 == n3 and n1 > 0
    for i in range(len(subs)):
        for key in p1[i]:
            _same(p1[i][key], p2[i][key], "pass 2, sub-batch %d, %s" % (i, key))
            _same(p1[i][key], p3[i][key], "no-recycle, sub-batch %d, %s" % (i, key))
        assert not p1[i]["status"].any()
    # integer checksums on every chunk
    for i, pk in enumerate(subs):
        o = p1[i]
        assert np.array_equal(o["occ_cov"], o["nuc_cov"] + o["nfr_cov"])
        l, n = pk.frag_lpos.astype(np.int64), pk.frag_ilen.astype(np.int64)
        cid = np.repeat(np.arange(pk.n_chunks), np.diff(pk.frag_off))
        r = l + n - 1
        want = np.bincount(cid[(l >= 0) & (l < TL)], minlength=pk.n_chunks) + np.bincount(cid[(r >= 0) & (r < TL)], minlength=pk.n_chunks)
        got = np.add.reduceat(o["ins"].astype(np.int64), pk.out_off[:-1])
        assert np.array_equal(got, want)
        assert np.array_equal(o["norm"], o["raw"] - o["bg"])
    # 60 of the 120 chunks against the oracle (every second chunk of every sub-batch)
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    picks = [(i, k) for i in range(len(subs)) for k in range(0, subs[i].n_chunks, 2)]
    tasks = []
    for i, k in picks:
        pk = subs[i]
        l, n = pk.chunk_frags(k)
        tasks.append((l, n, TL, pk.chunk_bias(k), pk.bias_left, par["vmat"], 105, 251, sizes, nucp, nfrp))
    with _spawn_pool() as pool:
        ref = pool.map(W.tracks_worker, tasks, chunksize=1)
    assert len(picks) >= 50
    for (i, k), want in zip(picks, ref):
        a, e = int(subs[i].out_off[k]), int(subs[i].out_off[k + 1])
        o = p2[i]                                            # the pass that ran on released batches
        for name in ("nuc_cov", "nfr_cov", "occ_cov"):
            assert np.array_equal(o[name][a:e], want[name]), name
        assert np.array_equal(o["ins"][a:e], want["ins"])
        for name in ("raw", "bg", "norm", "smoothed", "occ", "occ_lower", "occ_upper"):
            assert_track(o[name][a:e], want[name], "%s (sub-batch %d chunk %d)" % (name, i, k))


def test_auto_recycle_threshold(ctx):
    """ResidentShard's automatic decision: outputs are kept when they fit, recycled when all of them would not"""
    subs = _subs(2, 3)
    keep = ResidentShard(ctx, subs)                                   # 60 kbp: fits
    assert not keep.recycle
    keep.close()
    tight = ResidentShard(ctx, subs, mem_fraction=1e-6)               # pretend HBM is tiny
    assert tight.recycle
    stages = Stages(nuc_sd=10, occ=True, ins=None)
    tight.step(stages)
    with pytest.raises(L.NatacError):
        tight.batches[0].track(L.T_OCC)
    tight.close()
