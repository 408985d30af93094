"""Randomised sweep over round 4's device paths (development / release check), on the ragged chunk sets of fuzz_parity.py:
  * the stages of a batch in another order on a second context (ins, occ, nuc) == (nuc, occ, ins), bit for bit (round 4 ran the
    co-scheduled launch of DESIGN.md section 3.3c here; that path left the library in round 5);
  * natac_store_adopt / natac_store_read == the parse of the device's own text of the same tracks (what a reader of the file gets).
usage: python tests/fuzz/fuzz_round4.py [n_rounds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import golden  # noqa: E402
from nucleoatac_amd import _lib as L  # noqa: E402
from nucleoatac_amd.device import Context, TrackStore  # noqa: E402
from nucleoatac_amd.packing import PackedChunks  # noqa: E402
from nucleoatac_amd.synth import synth_occ_distributions, synth_size_distribution  # noqa: E402

TRACKS = (L.T_NUC_COV, L.T_NFR_COV, L.T_RAW, L.T_BACKGROUND, L.T_NORM, L.T_SMOOTH, L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV,
          L.T_INS, L.T_OCC_PREFILL)


def random_batch(rng):
    if rng.random() < 0.25:
        nch = int(rng.integers(40, 400))
        lens = [int(rng.integers(121, 900)) for _ in range(nch)]
    else:
        nch = int(rng.integers(1, 12))
        lens = [int(rng.choice([121, 122, 125, int(rng.integers(127, 700)), int(rng.integers(700, 3000)), int(rng.integers(3000, 9000))]))
                for _ in range(nch)]
    with_bias = rng.random() < 0.75
    fr = []
    for Lc in lens:
        dens = float(rng.choice([0.0, 0.02, 0.2, 1.0, 4.0]))
        nf = int(dens * Lc) if Lc < 3000 or dens <= 1.0 else Lc
        n = rng.integers(1, 420, size=nf)
        c = rng.integers(-200, Lc + 200, size=nf)
        if nf and rng.random() < 0.5:
            a, w = int(rng.integers(0, Lc)), int(rng.integers(50, 600))
            keep = (c < a - 60) | (c > a + w + 60)
            c, n = c[keep], n[keep]
        o = np.argsort(c, kind="stable")
        c, n = c[o], n[o]
        fr.append((c - (n - 1) // 2, n))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    boff = bias = None
    if with_bias:
        nb = [Lc + 493 for Lc in lens]
        bias = rng.normal(0, float(rng.choice([0.3, 0.8, 1.5])), size=sum(nb))
        if rng.random() < 0.3:           # a non-finite stretch: the FFT tiles fall back, NaNs stay confined
            a = int(rng.integers(0, len(bias) - 5))
            bias[a:a + int(rng.integers(1, 40))] = np.nan
        boff = np.concatenate(([0], np.cumsum(nb)))
    return PackedChunks(np.arange(nch) * 20000 + 10000, lens, off, np.concatenate([x[0] for x in fr]), np.concatenate([x[1] for x in fr]), boff, bias)


def everything(b, co, kw):
    if co:
        b.run_ins(0, 2000)
        b.run_occ()
        b.run_nuc(10)
    else:
        b.run_nuc(10)
        b.run_occ()
        b.run_ins(0, 2000)
    pk = b.run_peaks(**kw)
    out = [b.track(t) for t in TRACKS] + [b.grid(g) for g in (L.G_OCC, L.G_LOWER, L.G_UPPER)] + list(pk)
    out += list(b.run_occ_peaks(min_occ=0.1, sep=120)) + [b.status()]
    return out


def as_read_back(text, pk, chroms):
    out = np.full(int(pk.out_off[-1]), np.nan)
    where = {}
    for k in range(pk.n_chunks):
        where.setdefault(chroms[k], []).append((int(pk.chunk_start[k]), int(pk.chunk_start[k]) + int(pk.chunk_len[k]), int(pk.out_off[k])))
    for line in text.decode().splitlines():
        c, a, b, v = line.split("\t")
        a, b, v = int(a), int(b), float(v)
        for s, e, o in where[c]:
            if a >= s and b <= e:
                out[o + a - s:o + b - s] = v
                break
        else:
            raise AssertionError(line)
    return out


def make_ctx(env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        ctx = Context(0)
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
    par = golden("params_example")
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(synth_size_distribution(251))
    nucp, nfrp = synth_occ_distributions(251)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    return ctx


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    plain = make_ctx({})
    co = make_ctx({})
    store = TrackStore()
    t0 = time.time()
    bp = n_hard = 0
    for r in range(rounds):
        pk = random_batch(rng)
        kw = dict(min_signal=0, sep=int(rng.choice([25, 40, 120])), boundary=int(rng.choice([0, 30, 60])), order=int(rng.choice([1, 5, 12])))
        b0, b1 = plain.upload(pk), co.upload(pk)
        ref, got = everything(b0, False, kw), everything(b1, True, kw)
        for i, (x, y) in enumerate(zip(ref, got)):
            assert x.shape == y.shape and np.array_equal(x, y, equal_nan=x.dtype.kind == "f"), ("stage order changed a result", r, i)
        # the store: what a reader of the written track gets
        chroms = ["chr%d" % (1 + k % 3) for k in range(pk.n_chunks)]
        tracks = (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV)
        seg = store.adopt(b0, tracks)
        if seg is None:
            n_hard += 1
        else:
            for slot, t in enumerate(tracks):
                text, info = b0.format_track(t, chroms, pk.chunk_start, compress=False)
                want = as_read_back(text.tobytes(), pk, chroms)
                have = store.read(plain, [seg], [0], [pk.total_bp], slot)
                assert np.array_equal(have, want, equal_nan=True), ("store != file", r, t)
        b0.free()
        b1.free()
        bp += pk.total_bp
        if time.time() - t0 > float(os.environ.get("FUZZ_SECONDS", "1e9")):
            rounds = r + 1
            break
    store.close()
    plain.close()
    co.close()
    print("fuzz_round4 ok: %d rounds, %d bases, %d batches left to the file path by the store, %.0f s" % (rounds, bp, n_hard, time.time() - t0))


if __name__ == "__main__":
    main()
