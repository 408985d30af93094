"""GPU: device-side candidate search (natac_run_peaks) == utils.call_peaks applied to the same tracks on the host,
bit for bit (same jitter stream, same greedy thinning), plus statistics == natac_run_candidates on those positions."""
import numpy as np
import pytest

from helpers import golden, packed_from_golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.packing import PackedChunks
from nucleoatac_amd.synth import make_synthetic_chunks, synth_size_distribution

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from nucleoatac_amd.device import Context
    par = golden("params_example")
    c = Context(0)
    c.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    c.set_sizes(par["sizes"])
    yield c
    c.close()


def _host_peaks(b, pk, **kw):
    from oracle import natac_oracle as O
    norm, sm = b.split(b.track(L.T_NORM)), b.split(b.track(L.T_SMOOTH))
    cc, cp = [], []
    for k in range(pk.n_chunks):
        p = O.call_peaks((norm[k] + sm[k]).copy(), **kw)
        cc += [k] * len(p)
        cp += [int(x) for x in p]
    return np.array(cc, np.int32), np.array(cp, np.int32)


@pytest.mark.parametrize("case,with_bias", [("chunks_basic", True), ("chunks_gaps", True), ("chunks_nobias", False)])
def test_device_peaks_equal_host_call_peaks_on_golden(ctx, case, with_bias):
    g = golden(case)
    pk = packed_from_golden(g, with_bias)
    b = ctx.upload(pk)
    b.run_nuc(10)
    cc, cp, lr, var, z = b.run_peaks(min_signal=0, sep=25, boundary=60, order=12)
    hc, hp = _host_peaks(b, pk, min_signal=0, sep=25, boundary=60, order=12)
    assert np.array_equal(cc, hc) and np.array_equal(cp, hp)
    lr2, var2, z2 = b.run_candidates(cc, cp)
    assert np.array_equal(lr, lr2, equal_nan=True) and np.array_equal(var, var2, equal_nan=True) and np.array_equal(z, z2, equal_nan=True)
    # and they are the reference's candidates (those above its FFT noise floor)
    for k in range(pk.n_chunks):
        ref = g["c%d_cands" % k][:, 0].astype(np.int32)
        mine = cp[cc == k]
        assert set(ref) <= set(mine)
    assert not b.status().any()
    b.free()


def test_device_peaks_large_ragged_batch(ctx):
    rng = np.random.default_rng(3)
    pk = make_synthetic_chunks(3000, 2120, 500, seed=21)
    b = ctx.upload(pk)
    b.run_nuc(10)
    for kw in (dict(min_signal=0, sep=25, boundary=60, order=12), dict(min_signal=0.05, sep=120, boundary=30, order=1)):
        cc, cp, lr, var, z = b.run_peaks(**kw)
        hc, hp = _host_peaks(b, pk, **kw)
        assert len(cc) > 5000 and np.array_equal(cc, hc) and np.array_equal(cp, hp)
        assert np.all(np.diff(cc) >= 0)
    b.free()
    # ragged + an empty chunk + a chunk with a fragment-free stretch
    lens = [400, 2500, 1203, 777]
    fr, off = [], [0]
    for i, Lc in enumerate(lens):
        n = rng.integers(30, 300, size=0 if i == 0 else 3 * Lc // 10)
        c = rng.integers(-100, Lc + 100, size=len(n))
        if i == 1:
            c = c[(c < 800) | (c > 1300)]
            n = n[:len(c)]
        o = np.argsort(c, kind="stable")
        c, n = c[o], n[o]
        fr.append((c - (n - 1) // 2, n))
        off.append(off[-1] + len(n))
    pk2 = PackedChunks(np.arange(4) * 9000, lens, off, np.concatenate([x[0] for x in fr]), np.concatenate([x[1] for x in fr]), None, None)
    b = ctx.upload(pk2)
    b.run_nuc(10)
    cc, cp, lr, var, z = b.run_peaks()
    hc, hp = _host_peaks(b, pk2, min_signal=0, sep=25, boundary=60, order=12)
    assert np.array_equal(cc, hc) and np.array_equal(cp, hp) and (cc != 0).all()
    b.free()


@pytest.mark.parametrize("length,order", [(2048, 150), (4000, 150), (2120, 255), (3500, 97), (2120, 40)])
def test_device_peaks_with_a_large_order(ctx, length, order):
    """a large `order` (--redundant_sep up to 511) shrinks the LDS peak lists (pk_cap = maxL / (order + 1)) below the room the
    per-wave row masks of natac_peaks_chunk_reg need (maxL 4,096, order 150: 104 bytes for 128); the masks then stay in registers."""
    pk = make_synthetic_chunks(40, length, 600, seed=length + order)
    b = ctx.upload(pk)
    b.run_nuc(10)
    kw = dict(min_signal=0, sep=25, boundary=60, order=order)
    cc, cp, lr, var, z = b.run_peaks(**kw)
    hc, hp = _host_peaks(b, pk, **kw)
    assert len(cc) > 40 and np.array_equal(cc, hc) and np.array_equal(cp, hp)
    assert not b.status().any()
    b.free()


def test_run_peaks_argument_errors(ctx):
    pk = make_synthetic_chunks(2, 500, 50, seed=1)
    b = ctx.upload(pk)
    with pytest.raises(L.NatacError):
        b.run_peaks()                       # nuc stage has not run
    b.run_nuc(10)
    with pytest.raises(L.NatacError):
        b.run_peaks(order=0)
    b.free()


def test_track_peaks_on_occupancy_equal_host_call_peaks(ctx):
    """OccChunk.callPeaks parameters (sep 120, min_signal 0.1, order 1) on the NaN-filled smoothed occupancy track"""
    from oracle import natac_oracle as O
    from nucleoatac_amd.synth import synth_occ_distributions
    nucp, nfrp = synth_occ_distributions(251)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    pk = make_synthetic_chunks(2000, 1501, 300, seed=33)
    b = ctx.upload(pk)
    with pytest.raises(L.NatacError):
        b.run_track_peaks(L.T_OCC)          # stage has not run
    b.run_occ()
    cc, cp = b.run_track_peaks(L.T_OCC, min_signal=0.1, sep=120, boundary=60, order=1)
    occ = b.split(b.track(L.T_OCC))
    hc, hp = [], []
    for k in range(pk.n_chunks):
        p = O.call_peaks(occ[k].copy(), sep=120, min_signal=0.1)
        hc += [k] * len(p)
        hp += [int(x) for x in p]
    assert len(cc) > 3000 and np.array_equal(cc, hc) and np.array_equal(cp, hp)
    b.free()


def test_long_chunks_take_the_segmented_search_and_the_global_insertion_counts(ctx):
    """chunks longer than 4,096 bases: peak search in LDS segments (signal re-read per segment); longer than 15,360: insertion
    counts by global atomics instead of the per-chunk LDS histogram.  Both equal the host functions."""
    from oracle import natac_oracle as O
    rng = np.random.default_rng(17)
    lens = [17000, 4097, 2120]
    fr, off = [], [0]
    for Lc in lens:
        n = rng.integers(30, 400, size=Lc // 4)
        c = np.sort(rng.integers(-50, Lc + 50, size=len(n)))
        fr.append((c - (n - 1) // 2, n))
        off.append(off[-1] + len(n))
    pk = PackedChunks(np.arange(3) * 30000, lens, off, np.concatenate([x[0] for x in fr]), np.concatenate([x[1] for x in fr]), None, None)
    b = ctx.upload(pk)
    b.run_nuc(10)
    b.run_ins(0, 2000)
    for kw in (dict(min_signal=0, sep=25, boundary=60, order=12), dict(min_signal=0.02, sep=120, boundary=30, order=1)):
        cc, cp, lr, var, z = b.run_peaks(**kw)
        hc, hp = _host_peaks(b, pk, **kw)
        assert len(cc) > 100 and np.array_equal(cc, hc) and np.array_equal(cp, hp)
    ins = b.split(b.track(L.T_INS))
    for k, Lc in enumerate(lens):
        assert np.array_equal(ins[k], O.get_insertions(fr[k][0], fr[k][1], 0, Lc).astype(np.int32))
    assert not b.status().any()
    b.free()
