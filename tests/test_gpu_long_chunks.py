"""GPU: long merged chunks against the oracle.

ChunkList.merge (pyatac/chunk.py:109-125 of the reference) turns adjacent BED windows into ONE chunk; real peak sets give chunks of
tens to hundreds of kb.  One batch with chunks of 50,000, 200,000 and 1,000,003 bases next to ordinary ones:

* every float track of the nuc and the occ stage and the raw occupancy grid against `oracle.natac_oracle` on sampled ~5-kb windows of
  the long chunks (first, last and interior windows; the oracle is evaluated on the window as a chunk of its own with the long chunk's
  fragments and bias, which gives the long chunk's values except where a smoothing window crosses the sample's artificial edge);
* integer tracks (coverage, insertion counts) on every base, the candidate search against call_peaks on every chunk;
* the device writer (text == native host writer, members inflate) and the resident store (== what the text shows) on the same batch.
"""
import gzip
import io

import numpy as np
import pytest

from helpers import assert_track, cancel_scale, expand_grid, golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.packing import PackedChunks
from nucleoatac_amd.synth import synth_occ_distributions, synth_size_distribution

pytestmark = pytest.mark.gpu

LENS = [50000, 2120, 200000, 700, 1000003, 4097]
WIN = 5000


def _long_batch(seed=11):
    from nucleoatac_amd.synth import synth_centres, synth_sizes
    rng = np.random.default_rng(seed)
    fr = []
    for Lc in LENS:
        nf = int(Lc * 0.24)
        n = synth_sizes(rng, nf).astype(np.int64)
        c = np.sort(rng.integers(-200, Lc + 200, size=nf))
        if Lc >= 50000:        # two fragment-free stretches: NaN gaps in the occupancy tracks of a long chunk, one across a sampled window
            keep = ~(((c > 1200) & (c < 1700)) | ((c > Lc // 2 + 900) & (c < Lc // 2 + 1500)))
            c, n = c[keep], n[keep]
        fr.append((c - (n - 1) // 2, n))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    nb = [Lc + 493 for Lc in LENS]
    starts = np.concatenate(([0], np.cumsum(np.array(LENS[:-1]) + 3000))) + 100000
    return PackedChunks(starts, LENS, off, np.concatenate([x[0] for x in fr]), np.concatenate([x[1] for x in fr]),
                        np.concatenate(([0], np.cumsum(nb))), rng.normal(0, 0.6, size=sum(nb)))


def _windows(Lc):
    """sample windows [a, e) of a chunk: both ends and two interior ones; starts are multiples of 5 (the occupancy grid's phase)"""
    if Lc <= WIN + 1000:
        return [(0, Lc)]
    mid = (Lc // 2) // 5 * 5
    third = (Lc // 3) // 5 * 5 + 1000
    last = ((Lc - WIN) // 5) * 5
    return [(0, WIN), (mid, mid + WIN), (third, third + WIN), (last, Lc)]


@pytest.fixture(scope="module")
def setup():
    from nucleoatac_amd.device import Context
    par = golden("params_example")
    c = Context(0)
    sizes = synth_size_distribution(251)
    nucp, nfrp = synth_occ_distributions(251)
    c.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    c.set_sizes(sizes)
    c.set_occ_model(nucp, nfrp, step=5, flank=60)
    pk = _long_batch()
    b = c.upload(pk)
    b.run_nuc(10)
    b.run_occ()
    b.run_ins(0, 2000)
    yield c, b, pk, par, sizes, nucp, nfrp
    b.free()
    c.close()


def test_float_tracks_and_grid_match_the_oracle_on_sampled_windows(setup):
    from oracle import natac_oracle as O
    ctx, b, pk, par, sizes, nucp, nfrp = setup
    assert not b.status().any()
    ids = (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH, L.T_OCC_PREFILL, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV)
    tr = {t: b.split(b.track(t)) for t in ids}
    grids = [b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)]
    goff = np.concatenate(([0], np.cumsum([len(range(2, Lc, 5)) for Lc in LENS])))
    assert goff[-1] == b.grid_info()[1]
    n_win = n_nan = 0
    for k, Lc in enumerate(LENS):
        l, n = pk.chunk_frags(k)
        l, n = l.astype(np.int64), n.astype(np.int64)
        cen = l + (n - 1) // 2
        bias = pk.chunk_bias(k)
        gk = [expand_grid(g[goff[k]:goff[k + 1]], Lc) for g in grids]
        for a, e in _windows(Lc):
            m = (cen >= a - 700) & (cen < e + 700)
            nt = O.nuc_chunk_tracks(l[m], n[m], a, e, bias, -pk.bias_left, par["vmat"], 105, 251, sizes)
            oc = O.occ_chunk_tracks(l[m], n[m], a, e, bias, -pk.bias_left, nucp, nfrp)
            sl = slice(a, e)
            assert_track(tr[L.T_NUC_COV][k][sl], nt["nuc_cov"], "nuc_cov", exact=True)
            assert_track(tr[L.T_NFR_COV][k][sl], nt["nfr_cov"], "nfr_cov", exact=True)
            assert_track(tr[L.T_OCC_COV][k][sl], oc["cov"], "occ cov", exact=True)
            assert_track(tr[L.T_RAW][k][sl], nt["raw"], "raw")
            assert_track(tr[L.T_BACKGROUND][k][sl], nt["bg"], "background")
            assert_track(tr[L.T_NORM][k][sl], nt["norm"], "norm", scale=cancel_scale(nt["raw"], nt["bg"]))
            # raw occupancy grid: discrete alpha values, bit-exact; the sample's last grid block may be cut by ITS end, not the chunk's
            ee = e if e == Lc else e - 5
            for got, name in zip(gk, ("occ", "occ_lower", "occ_upper")):
                assert_track(got[a:ee], oc[name][:ee - a], "grid " + name, exact=True)
            # smoothed tracks: away from the sample's artificial edges (half windows: 30 nuc, 60 occ + the cut grid block)
            ia, ie = (0 if a == 0 else 35), (e - a if e == Lc else e - a - 35)
            assert_track(tr[L.T_SMOOTH][k][a + ia:a + ie], nt["smoothed"][ia:ie], "smoothed", scale=cancel_scale(nt["raw"], nt["bg"]))
            ia, ie = (0 if a == 0 else 70), (e - a if e == Lc else e - a - 70)
            for t, name in ((L.T_OCC_PREFILL, "smoothed_vals"), (L.T_OCC_LOWER, "smoothed_lower"), (L.T_OCC_UPPER, "smoothed_upper")):
                assert_track(tr[t][k][a + ia:a + ie], oc[name][ia:ie], name)
            n_nan += int(np.isnan(oc["smoothed_lower"][ia:ie]).sum())
            n_win += 1
    assert n_win == 3 * 4 + 3 and n_nan > 100           # the NaN gaps of the long chunks were inside sampled windows


def test_integer_tracks_and_candidates_on_every_base(setup):
    from oracle import natac_oracle as O
    ctx, b, pk, par, sizes, nucp, nfrp = setup
    ins = b.split(b.track(L.T_INS))
    cov = b.split(b.track(L.T_OCC_COV))
    nuc_cov, nfr_cov = b.split(b.track(L.T_NUC_COV)), b.split(b.track(L.T_NFR_COV))
    for k, Lc in enumerate(LENS):
        l, n = pk.chunk_frags(k)
        l, n = l.astype(np.int64), n.astype(np.int64)
        assert np.array_equal(ins[k], O.get_insertions(l, n, 0, Lc).astype(np.int32)), k
        # coverage = 121-wide box sum of the per-base centre counts (pyatac/tracks.py:209-222), every base of the long chunks
        cen = l + (n - 1) // 2
        ok = (n >= 0) & (n < 251)
        cnt = np.bincount(np.clip(cen[ok] + 60, 0, Lc + 120)[(cen[ok] >= -60) & (cen[ok] < Lc + 60)], minlength=Lc + 121)[:Lc + 120]
        box = np.convolve(cnt, np.ones(121, dtype=np.int64), mode="valid")
        assert np.array_equal(cov[k], box.astype(np.float64)), k
        assert np.array_equal(nuc_cov[k] + nfr_cov[k], cov[k]), k
    # the fill of call_peaks (chunk minimum into the NaN gaps) on a long chunk
    occ, pre = b.split(b.track(L.T_OCC)), b.split(b.track(L.T_OCC_PREFILL))
    for k in (0, 2, 4):
        gap = np.isnan(pre[k])
        assert gap.sum() > 300 and not np.isnan(occ[k]).any()
        assert np.array_equal(occ[k][~gap], pre[k][~gap]) and np.all(occ[k][gap] == np.nanmin(pre[k]))
    norm, sm = b.split(b.track(L.T_NORM)), b.split(b.track(L.T_SMOOTH))
    for kw in (dict(min_signal=0, sep=25, boundary=60, order=12), dict(min_signal=0.05, sep=120, boundary=30, order=1)):
        cc, cp, lr, var, z = b.run_peaks(**kw)
        hc, hp = [], []
        for k in range(pk.n_chunks):
            p = O.call_peaks((norm[k] + sm[k]).copy(), **kw)
            hc += [k] * len(p)
            hp += [int(x) for x in p]
        assert len(cc) > 1000 and np.array_equal(cc, np.array(hc, np.int32)) and np.array_equal(cp, np.array(hp, np.int32))
        lr2, var2, z2 = b.run_candidates(cc, cp)
        assert np.array_equal(lr, lr2, equal_nan=True) and np.array_equal(var, var2, equal_nan=True) and np.array_equal(z, z2, equal_nan=True)
    # lr / var / z of the candidates inside one sampled window of the 1,000,003-base chunk against the oracle's dense matrices
    cc, cp, lr, var, z = b.run_peaks(min_signal=0, sep=25, boundary=60, order=12)
    k, a, e = 4, 500000, 505000
    l, n = pk.chunk_frags(k)
    l, n = l.astype(np.int64), n.astype(np.int64)
    cen = l + (n - 1) // 2
    m = (cen >= a - 700) & (cen < e + 700)
    nt = O.nuc_chunk_tracks(l[m], n[m], a, e, pk.chunk_bias(k), -pk.bias_left, par["vmat"], 105, 251, sizes)
    sel = np.flatnonzero((cc == k) & (cp >= a + 130) & (cp < e - 130))
    assert len(sel) > 50
    checked = 0
    for j in sel[::3]:
        p = int(cp[j]) - a
        if nt["nuc_cov"][p] < 1:
            continue
        ref_lr = O.get_lr(nt["mat"], nt["mat_start"], nt["bmat"], nt["b0"], nt["b_start"], par["vmat"], 105, 251, p + a)
        pr = O.signal_distribution_probs(nt["bmat"], nt["b_start"], 105, 251, 60, p + a)
        ref_z, ref_var = O.z_score(nt["norm"][p], nt["nuc_cov"][p], pr, par["vmat"])
        assert abs(lr[j] - ref_lr) <= 1e-5 * abs(ref_lr) + 1e-7 and abs(var[j] - ref_var) <= 1e-5 * abs(ref_var) + 1e-12
        assert abs(z[j] - ref_z) <= 1e-5 * abs(ref_z) + 1e-7
        checked += 1
    assert checked > 10


def test_device_writer_and_resident_store_on_the_long_batch(setup, tmp_path):
    from nucleoatac_amd.device import TrackStore
    from nucleoatac_amd.writer import BGZF_EOF, bgzf_lines_host, write_bedgraph
    ctx, b, pk, par, sizes, nucp, nfrp = setup
    chroms = ["chr1", "chr1", "chr1", "chr2", "chr2", "chrUn_long_name"]
    for t in (L.T_OCC, L.T_OCC_LOWER, L.T_SMOOTH, L.T_INS):
        vals = b.track(t).astype(np.float64)
        p = str(tmp_path / "native.bedgraph")
        write_bedgraph(p, chroms, pk.chunk_start, pk.out_off, vals, compress=0)
        want = open(p, "rb").read()
        text, info = b.format_track(t, chroms, pk.chunk_start, compress=False)
        assert info["hard"] == 0 and text.tobytes() == want, t
        z, zi = b.format_track(t, chroms, pk.chunk_start, compress=True)
        assert gzip.GzipFile(fileobj=io.BytesIO(z.tobytes() + BGZF_EOF)).read() == want, t
        assert z.tobytes() == bgzf_lines_host(want), t
    store = TrackStore()
    tracks = (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER)
    seg = store.adopt(b, tracks)
    assert seg is not None
    for slot, t in enumerate(tracks):
        text, _ = b.format_track(t, chroms, pk.chunk_start, compress=False)
        # what a reader of the text gets for the 1,000,003-base chunk: value per base, NaN where no line covers it
        k = 4
        s0, off = int(pk.chunk_start[k]), int(pk.out_off[k])
        want = np.full(LENS[k], np.nan)
        for line in text.tobytes().decode().splitlines():
            c, a, e, v = line.split("\t")
            if c == chroms[k] and int(a) >= s0 and int(e) <= s0 + LENS[k]:
                want[int(a) - s0:int(e) - s0] = float(v)
        got = store.read(ctx, [seg], [off], [LENS[k]], slot)
        assert np.array_equal(got, want, equal_nan=True), t
    store.close()
