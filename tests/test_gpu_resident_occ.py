"""GPU: the occupancy tracks kept in HBM between the steps of `nucleoatac run` (nucleoatac_amd/occstore.py, natac_store_*).

1. natac_store_adopt holds, per base, exactly what a reader of the written track gets: the device text of the same track is
   parsed back on the host (float of every line's value, NaN where no line covers a base) and compared bit for bit -- on the
   golden batches with fragment-free stretches (NaN runs, runs lost before a NaN, exact zeros) and on a synthetic batch.
2. `nucleoatac run` with the resident tracks is byte-identical, file for file, to the run that reads the occupancy tracks back
   from the files it wrote (the reference's way, cli.py:34-64) -- and the resident run reads no occupancy text at all."""
import gzip
import os

import numpy as np
import pytest

from helpers import GOLDEN, golden, packed_from_golden, read_bed3, synth_saccer3
from nucleoatac_amd import _lib as L
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution

pytestmark = pytest.mark.gpu


def _ctx():
    from nucleoatac_amd.device import Context
    ctx = Context(0)
    par = golden("params_example")
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(synth_size_distribution(251))
    nucp, nfrp = synth_occ_distributions(251)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    return ctx


def _as_read_back(text, pk, chroms):
    """per-base values a reader of this bedGraph text gets (Track.read_track with empty = nan; later records win)"""
    out = np.full(int(pk.out_off[-1]), np.nan)
    start_of = {}
    for k in range(pk.n_chunks):
        start_of.setdefault(chroms[k], []).append((int(pk.chunk_start[k]), int(pk.chunk_start[k]) + int(pk.chunk_len[k]), int(pk.out_off[k])))
    for line in text.decode().splitlines():
        c, a, b, v = line.split("\t")
        a, b, v = int(a), int(b), float(v)
        for s, e, off in start_of[c]:
            if a >= s and b <= e:
                out[off + a - s:off + b - s] = v
                break
        else:
            raise AssertionError("line outside every chunk: " + line)
    return out


@pytest.mark.parametrize("case", ["chunks_gaps", "chunks_basic", "synthetic"])
def test_adopted_tracks_are_what_the_file_shows(case):
    from nucleoatac_amd.device import TrackStore
    if case == "synthetic":
        counts = np.full(60, 300, dtype=np.int64)
        counts[::9] = 0                                        # fragment-free chunks: all-NaN occupancy
        counts[1::9] = 3                                       # nearly empty: long exact-zero / NaN stretches
        pk = make_synthetic_chunks(60, 1500, 300, seed=21, counts=counts)
    else:
        pk = packed_from_golden(golden(case))
    chroms = ["chr%d" % (1 + k % 3) for k in range(pk.n_chunks)]
    with _ctx() as ctx:
        b = ctx.upload(pk)
        b.run_occ()
        b.run_nuc(10)
        store = TrackStore()
        tracks = (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_OCC_COV)
        seg = store.adopt(b, tracks)
        assert seg is not None and store.info()["bytes"] == 4 * 8 * pk.total_bp
        n_nan = 0
        for slot, t in enumerate(tracks):
            text, info = b.format_track(t, chroms, pk.chunk_start, compress=False)
            assert not info["hard"]
            want = _as_read_back(text.tobytes(), pk, chroms)
            got = store.read(ctx, [seg], [0], [pk.total_bp], slot)
            assert np.array_equal(got, want, equal_nan=True), (case, t)
            raw = b.track(t)
            assert np.array_equal(np.isnan(got) | (got == raw) | (np.abs(got - raw) <= 1e-11 * np.abs(raw)), np.ones(len(got), bool))
            n_nan += int(np.isnan(got).sum())
            # ranges and single positions
            rng = np.random.default_rng(slot)
            off = rng.integers(0, pk.total_bp - 50, size=200)
            ln = rng.integers(1, 50, size=200)
            part = store.read(ctx, [seg] * 200, off, ln, slot)
            assert np.array_equal(part, np.concatenate([want[o:o + l] for o, l in zip(off, ln)]), equal_nan=True)
        if case != "chunks_basic":
            assert n_nan > 0
        # a rejected request leaves the store usable
        with pytest.raises(Exception):
            store.read(ctx, [seg], [pk.total_bp - 5], [10], 0)
        with pytest.raises(Exception):
            store.read(ctx, [seg + 1], [0], [1], 0)
        assert len(store.read(ctx, [seg], [3], [4], 1)) == 4
        store.close()
        b.free()


def test_run_with_resident_tracks_is_byte_identical_to_the_file_path(tmp_path, monkeypatch):
    from nucleoatac_amd import occstore
    from nucleoatac_amd.nucleoatac import NucleosomeCalling as NC
    from nucleoatac_amd.nucleoatac.cli import main
    bed = os.path.join(GOLDEN, "ref_example.bed")
    bam, fa = synth_saccer3(str(tmp_path), read_bed3(bed), seed=3)
    reads = {"files": 0}
    real = NC.read_regions_of

    def counting(path, chunks, value_col=4):
        if ".occ." in os.path.basename(path):
            reads["files"] += 1
        return real(path, chunks, value_col)

    monkeypatch.setattr(NC, "read_regions_of", counting)
    outs = {}
    for mode in (True, False):
        monkeypatch.setattr(occstore, "ENABLED", mode)
        reads["files"] = 0
        out = str(tmp_path / ("resident" if mode else "files"))
        main(["run", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--write_all", "--cores", "4"])
        outs[mode] = (out, reads["files"])
        assert occstore.lookup(out + ".occ.bedgraph.gz") is None          # released at the end of `run`
    assert outs[True][1] == 0 and outs[False][1] > 0                      # the resident run parsed no occupancy text
    for suffix in ("occ.bedgraph.gz", "occ.lower_bound.bedgraph.gz", "occ.upper_bound.bedgraph.gz", "occpeaks.bed.gz", "nucpos.bed.gz",
                   "nucpos.redundant.bed.gz", "nucleoatac_signal.bedgraph.gz", "nucleoatac_signal.smooth.bedgraph.gz",
                   "nucmap_combined.bed.gz", "nfrpos.bed.gz", "ins.bedgraph.gz"):
        a = gzip.open(outs[True][0] + "." + suffix, "rb").read()
        b = gzip.open(outs[False][0] + "." + suffix, "rb").read()
        assert a == b, suffix
        assert len(a) > 0 or suffix == "nfrpos.bed.gz"


def test_store_budget_declines_and_the_run_finishes_through_the_files(tmp_path, monkeypatch):
    """the store must never be what fills HBM.  A segment beyond the byte budget, or one that would leave the device less than
    min_free, is declined (None, counted, the store closes itself); `nucleoatac run` with a store that keeps nothing finishes through the
    files, byte-identical to the resident run."""
    from nucleoatac_amd import occstore
    from nucleoatac_amd.device import TrackStore
    from nucleoatac_amd.nucleoatac.cli import main
    pk = make_synthetic_chunks(20, 1500, 300, seed=4)
    tracks = (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER)
    with _ctx() as ctx:
        b = ctx.upload(pk)
        b.run_occ()
        store = TrackStore()
        need = 3 * 8 * pk.total_bp
        store.set_budget(max_bytes=2 * need - 1)
        assert store.adopt(b, tracks) == 0 and store.info() == dict(segments=1, bytes=need, declined=0)
        assert store.adopt(b, tracks) is None and store.info()["declined"] == 1        # the second one would pass the cap
        store.set_budget(max_bytes=-1)                                                  # an explicit budget re-opens the store
        assert store.adopt(b, tracks) == 1
        store.set_budget(min_free_bytes=1 << 50)                                        # more headroom than any device has
        assert store.adopt(b, tracks) is None and store.adopt(b, tracks) is None and store.info()["declined"] == 3
        assert store.info()["segments"] == 2 and len(store.read(ctx, [1], [3], [4], 1)) == 4
        store.close()
        b.free()
    bed = os.path.join(GOLDEN, "ref_example.bed")
    bam, fa = synth_saccer3(str(tmp_path), read_bed3(bed), seed=3)
    outs = {}
    for mode in ("resident", "declined"):
        if mode == "declined":
            monkeypatch.setenv("NATAC_STORE_MAX_BYTES", "1000")
        out = str(tmp_path / mode)
        main(["run", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out, "--cores", "4"])
        outs[mode] = out
    for suffix in ("occ.bedgraph.gz", "nucpos.bed.gz", "nucmap_combined.bed.gz", "nfrpos.bed.gz", "nucleoatac_signal.smooth.bedgraph.gz"):
        assert gzip.open(outs["resident"] + "." + suffix, "rb").read() == gzip.open(outs["declined"] + "." + suffix, "rb").read(), suffix
