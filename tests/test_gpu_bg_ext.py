"""Extended tiles of the FFT background kernel (natac_fft_bg.hpp: 16 more outputs on each side of a 512-point tile, finished by the
edge pass at the end of the tile's wave) against direct summation, the plain tiling and the oracle (NucleosomeCalling.py:49-64)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from helpers import golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.packing import PackedChunks
from nucleoatac_amd.synth import synth_size_distribution

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# lengths on both sides of the rule: every tile extended (848, 1272, 2120), extended tiles followed by plain ones with a cut last tile
# (800, 1200, 2000, 4100), one extended tile without a right edge (393), plain tiles only (61, 392, 440, 857, 1704, 3000)
LENGTHS = [2120, 848, 800, 1272, 1200, 2000, 440, 1704, 4100, 857, 61, 392, 3000, 2120, 393]


def _ragged_batch(lens, seed, bias_kind="normal"):
    rng = np.random.default_rng(seed)
    fr = []
    for Lc in lens:
        m = 40 + Lc // 4
        n = rng.integers(30, 300, size=m)
        l = rng.integers(-150, Lc + 100, size=m)
        o = np.argsort(l + (n - 1) // 2, kind="stable")
        fr.append((l[o], n[o]))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    nb = [Lc + 493 for Lc in lens]
    boff = np.concatenate(([0], np.cumsum(nb)))
    bias = None
    if bias_kind != "none":
        bias = rng.normal(0, 0.7, size=sum(nb))
    if bias_kind == "damaged":
        # chunk 0 (2,120 bases, extended): a NaN next to a tile border (base 424 + 3), chunk 1 (848): exp(-inf) = 0 inside its second tile,
        # chunk 8 (4,100): a dynamic range of e^25 across the border of tiles 2 and 3 -- the tiles whose (extended) windows hold them go
        # to direct summation, edge outputs included
        bias[boff[0] + 246 + 427] = np.nan
        bias[boff[1] + 246 + 600] = -np.inf
        bias[boff[8] + 246 + 1250:boff[8] + 246 + 1300] += 25.0
    pk = PackedChunks(np.arange(len(lens)) * 20000, lens, off, np.concatenate([x[0] for x in fr]),
                      np.concatenate([x[1] for x in fr]), boff, bias)
    return pk, fr


CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from helpers import golden
from test_gpu_bg_ext import _ragged_batch, LENGTHS
from nucleoatac_amd import _lib as L
from nucleoatac_amd.device import Context
from nucleoatac_amd.synth import synth_size_distribution
par = golden("params_example")
c = Context(0); c.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"])); c.set_sizes(synth_size_distribution(251))
pk, fr = _ragged_batch(LENGTHS, 7, sys.argv[2])
b = c.upload(pk); b.run_nuc(10)
cc = np.concatenate([np.full(len(range(70, Lc - 70, 3)), k) for k, Lc in enumerate(LENGTHS)]).astype(np.int32)
cp = np.concatenate([np.arange(70, Lc - 70, 3) for Lc in LENGTHS]).astype(np.int32)
lr, var, z = b.run_candidates(cc, cp)
np.savez(sys.argv[1], tracks=np.stack([b.track(L.T_BACKGROUND), b.track(L.T_NORM), b.track(L.T_SMOOTH)]), lr=lr, var=var, z=z,
         tiling=np.array([c.bg_tiling(Lc) for Lc in LENGTHS]))
b.free(); c.close()
''' % (ROOT, HERE)


def _run(mode_env, bias_kind, td, tag):
    env = dict(os.environ)
    env.pop("NATAC_BG_DIRECT", None)
    env.pop("NATAC_BG_EXT", None)
    env.update(mode_env)
    path = os.path.join(td, tag + ".npz")
    subprocess.run([sys.executable, "-c", CODE, path, bias_kind], check=True, env=env)
    return np.load(path)


def test_tiling_rule():
    """(tiles, extended tiles among them): the cheapest mix at 11 % extra per extended tile; a function of the length alone"""
    from nucleoatac_amd.device import Context
    par = golden("params_example")
    ctx = Context(0)
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    want = {2120: (5, 5), 848: (2, 2), 800: (2, 1), 1272: (3, 3), 1200: (3, 1), 2000: (5, 2), 4100: (10, 6), 393: (1, 1), 9900: (25, 4),
            857: (3, 0), 440: (2, 0), 1704: (5, 0), 61: (1, 0), 392: (1, 0), 3000: (8, 0), 10120: (26, 0), 1000003: (2551, 1)}
    got = {Lc: ctx.bg_tiling(Lc) for Lc in want}
    ctx.close()
    assert got == want


@pytest.mark.parametrize("bias_kind", ["normal", "none"])
def test_extended_tiles_equal_direct_summation_and_plain_tiles(bias_kind):
    """background, normalised and smoothed signal and the candidates' lr / var / z (which read the window sums the background stage
    leaves per base) of a ragged batch: extended tiles vs the direct-summation kernel vs plain tiles, ~1e-13"""
    with tempfile.TemporaryDirectory() as td:
        ext = _run({}, bias_kind, td, "ext")
        direct = _run({"NATAC_BG_DIRECT": "1"}, bias_kind, td, "direct")
        plain = _run({"NATAC_BG_EXT": "0"}, bias_kind, td, "plain")
    assert ext["tiling"][:, 1].sum() >= 8 and plain["tiling"][:, 1].sum() == 0 and direct["tiling"][:, 0].sum() == 0
    for other in (direct, plain):
        for t in range(3):
            np.testing.assert_allclose(ext["tracks"][t], other["tracks"][t], rtol=1e-10, atol=1e-12)
        for k in ("lr", "var", "z"):
            np.testing.assert_allclose(ext[k], other[k], rtol=1e-9, atol=1e-11)
    assert not np.array_equal(ext["tracks"][0], plain["tracks"][0])          # really two tilings
    # bases the edge pass finishes (chunk 0: 2,120 bases, tiles of 424): the outputs on both sides of every border
    o = 0
    edge = np.concatenate([np.arange(424 * k - 16, 424 * k + 16) for k in range(1, 5)] + [np.arange(0, 16), np.arange(2104, 2120)])
    np.testing.assert_allclose(ext["tracks"][0][o + edge], direct["tracks"][0][o + edge], rtol=1e-11, atol=0)


def test_extended_tiles_with_damaged_bias_match_oracle_and_direct():
    """a NaN, a zero and a huge dynamic range next to the borders of extended tiles: the tiles whose windows hold them are evaluated by
    direct summation, edge outputs included; NaNs sit exactly where the reference's dense correlation has them"""
    from oracle import natac_oracle as O
    par = golden("params_example")
    sizes = synth_size_distribution(251)
    with tempfile.TemporaryDirectory() as td:
        ext = _run({}, "damaged", td, "ext")
        direct = _run({"NATAC_BG_DIRECT": "1"}, "damaged", td, "direct")
    bg_e, bg_d = ext["tracks"][0], direct["tracks"][0]
    assert np.array_equal(np.isnan(bg_e), np.isnan(bg_d))
    ok = ~np.isnan(bg_d)
    np.testing.assert_allclose(bg_e[ok], bg_d[ok], rtol=1e-9, atol=1e-12)
    pk, fr = _ragged_batch(LENGTHS, 7, "damaged")
    off = np.concatenate(([0], np.cumsum(LENGTHS)))
    for k in (0, 1, 8):
        l, n = fr[k]
        with np.errstate(all="ignore"):
            nt = O.nuc_chunk_tracks(l.astype(np.int64), n.astype(np.int64), 0, LENGTHS[k], pk.chunk_bias(k), -246, par["vmat"], 105, 251, sizes)
        mine = bg_e[off[k]:off[k + 1]]
        assert np.array_equal(np.isnan(mine), np.isnan(nt["bg"])), k
        good = ~np.isnan(mine)
        # north-star tolerance here on purpose: next to e^25 entries the ORACLE's dense FFT correlation (scipy, as the reference's) carries
        # an error relative to the window's largest product into the small outputs; the kernel's own check is the 1e-9 comparison with
        # direct summation above
        np.testing.assert_allclose(mine[good], nt["bg"][good], rtol=1e-5, atol=1e-9)
    assert 0 < np.isnan(bg_e[:2120]).sum() < 400


CODE_GEOM = r'''
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from test_gpu_bg_ext import _ragged_batch, LENGTHS
from nucleoatac_amd import _lib as L
from nucleoatac_amd.device import Context
from nucleoatac_amd.synth import synth_size_distribution
lo, up = int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(lo * 1000 + up)
vm = rng.random((up - lo, 121)) * 0.01 + 1e-4
c = Context(0); c.set_vmat(vm, lo, up); c.set_sizes(synth_size_distribution(up))
pk, fr = _ragged_batch(LENGTHS, 11, "normal")
b = c.upload(pk); b.run_nuc(10)
np.savez(sys.argv[1], tracks=np.stack([b.track(L.T_BACKGROUND), b.track(L.T_NORM), b.track(L.T_RAW)]))
b.free(); c.close()
''' % (ROOT, HERE)


@pytest.mark.parametrize("lower,upper", [(105, 251), (104, 250), (105, 253), (104, 252), (105, 252), (106, 251), (61, 121)])
def test_pair_loop_variants_equal_direct_summation(lower, upper):
    """round 6: the skewed pair loop of natac_background_fft for an odd and an even first insert size (they differ in the ADDRESSES of the
    two operand streams only), an odd and an even number of row pairs (the loop is unrolled by two: 74 pairs leave one trip over), and an
    odd row count (the plain loop) -- each against the direct-summation kernel on the ragged batch, extended and plain tiles mixed."""
    def run(env_extra, tag, td):
        env = dict(os.environ)
        env.pop("NATAC_BG_DIRECT", None)
        env.pop("NATAC_BG_EXT", None)
        env.update(env_extra)
        path = os.path.join(td, tag + ".npz")
        subprocess.run([sys.executable, "-c", CODE_GEOM, path, str(lower), str(upper)], check=True, env=env)
        return np.load(path)["tracks"]
    with tempfile.TemporaryDirectory() as td:
        fft = run({}, "fft", td)
        direct = run({"NATAC_BG_DIRECT": "1"}, "direct", td)
    assert np.array_equal(fft[2], direct[2])                                  # raw does not go through the FFT
    assert np.isfinite(direct[0]).all() and np.abs(direct[0]).max() > 0
    np.testing.assert_allclose(fft[0], direct[0], rtol=1e-10, atol=1e-13)      # background
    np.testing.assert_allclose(fft[1], direct[1], rtol=1e-10, atol=1e-12)      # norm = raw - bg
