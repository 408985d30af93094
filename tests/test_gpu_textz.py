"""GPU: the device-side track writer (natac_batch_format_track: csrc/natac_textfmt.hpp + natac_textz.hpp + natac_deflate.hpp).
Text: byte-identical to the native host writer (natac_write_bedgraph, itself pinned to the reference's Track.write_track rows and
python-2 float formatting by tests/test_writer.py).  BGZF: inflates to that text and equals the host restatement of the encoder
byte for byte; the file is a valid tabix-indexable bedGraph."""
import gzip
import io
import os

import numpy as np
import pytest

from helpers import golden, packed_from_golden
from nucleoatac_amd import _lib as L
from nucleoatac_amd.packing import PackedChunks
from nucleoatac_amd.pyatac.tracks import _py2_float_str as f2s
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution
from nucleoatac_amd.writer import BGZF_EOF, bgzf_lines_host, tabix_index, write_bedgraph

pytestmark = pytest.mark.gpu

TRACKS = (L.T_NORM, L.T_SMOOTH, L.T_RAW, L.T_BACKGROUND, L.T_OCC, L.T_OCC_PREFILL, L.T_OCC_LOWER, L.T_OCC_UPPER, L.T_NUC_COV, L.T_INS)


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


def test_device_formatter_is_python2_str(ctx):
    rng = np.random.default_rng(0)
    vals = np.concatenate([
        rng.random(100000), rng.normal(0, 1, 100000) * np.exp(rng.normal(0, 8, 100000)), 10.0 ** rng.uniform(-320, 308, 50000),
        rng.integers(0, 10 ** 6, 50000).astype(float), (rng.integers(0, 10 ** 13, 50000) + 0.5) / 10.0 ** rng.integers(0, 14, 50000),
        np.frombuffer(rng.integers(0, 2 ** 63, 100000, dtype=np.int64).tobytes(), dtype=np.float64),
        np.array([0.0, -0.0, 1.0, 0.1, 1e-4, 9.99999999999e-5, 1e11, 1e12, 999999999999.5, 999999999999.4999, 1e-5, 123456789012.5,
                  5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf, -np.inf, 1e22, 0.30000000000000004, 1e-44, 1e-45,
                  100000000000.5, 99999999999.95, 2.5, 0.125])])
    vals = vals[~np.isnan(vals)]
    got, hard = ctx.format_doubles(vals)
    assert hard == 0                      # undecidable cases need an exact tie at |v| >= 1e12: none in this sample
    want = [f2s(float(v)) for v in vals]
    bad = [(v, g, w) for v, g, w in zip(vals, got, want) if g != w]
    assert not bad, bad[:10]
    # the only inputs the truncated power-of-ten table cannot decide: exact ties beyond its exact range -- counted, not guessed
    _, hard = ctx.format_doubles(np.array([1234567890125.0, 875485468971500.0]))
    assert hard == 2


def _native_text(tmp_path, chroms, starts, out_off, vals, **kw):
    p = str(tmp_path / "native.bedgraph")
    write_bedgraph(p, chroms, starts, out_off, vals, compress=0, **kw)
    return open(p, "rb").read()


def _ragged_batch(ctx):
    rng = np.random.default_rng(5)
    lens = [121, 122, 333, 1204, 700, 2120, 640]
    fr = []
    for i, Lc in enumerate(lens):
        if i == 2:
            l, n = np.zeros(0, np.int64), np.zeros(0, np.int64)
        else:
            n = np.concatenate((rng.integers(1, 400, size=60 + i * 40), [1, 2, 0, 1999, 2500, 250, 251]))
            l = rng.integers(-300, Lc + 200, size=len(n))
            if i == 5:                                   # a fragment-free stretch: NaN gaps in the occupancy tracks
                keep = (l + n // 2 < 600) | (l + n // 2 > 1500)
                l, n = l[keep], n[keep]
        o = np.argsort(l + (n - 1) // 2, kind="stable")
        fr.append((l[o], n[o]))
    off = np.concatenate(([0], np.cumsum([len(x[0]) for x in fr])))
    nb = [Lc + 493 for Lc in lens]
    pk = PackedChunks(np.arange(len(lens)) * 5000 + 999000, lens, off, np.concatenate([x[0] for x in fr]),
                      np.concatenate([x[1] for x in fr]), np.concatenate(([0], np.cumsum(nb))), rng.normal(0, 0.7, size=sum(nb)))
    chroms = ["chrI", "chrI", "chrII", "chrII", "scaffold_12_random", "chrX", "chrX"]
    return pk, chroms


def test_device_text_equals_native_writer(ctx, tmp_path):
    cases = []
    pk, chroms = _ragged_batch(ctx)
    cases.append((pk, chroms))
    for name in ("chunks_basic", "chunks_gaps"):
        g = golden(name)
        gp = packed_from_golden(g)
        cases.append((gp, ["chr%d" % (k % 3 + 1) for k in range(gp.n_chunks)]))
    n_lines = n_nan_runs = 0
    for pk, chroms in cases:
        b = ctx.upload(pk)
        b.run_nuc(10)
        b.run_occ()
        b.run_ins(0, 2000)
        for t in TRACKS:
            vals = b.track(t).astype(np.float64)
            for wz, keep in ((True, False), (False, False), (True, True)):
                text, info = b.format_track(t, chroms, pk.chunk_start, write_zero=wz, keep_runs_before_nan=keep, compress=False)
                want = _native_text(tmp_path, chroms, pk.chunk_start, pk.out_off, vals, write_zero=wz, keep_runs_before_nan=keep)
                assert info["hard"] == 0
                assert text.tobytes() == want, (t, wz, keep)
                assert info["text_bytes"] == len(want) and info["lines"] == want.count(b"\n")
                n_lines += info["lines"]
            n_nan_runs += int(np.isnan(vals).any())
        b.free()
    assert n_lines > 100000 and n_nan_runs >= 3           # NaN runs (and the runs before them) were exercised


def test_device_bgzf_members(ctx, tmp_path):
    from nucleoatac_amd.pyatac.tracks import Track
    pk = make_synthetic_chunks(300, 2120, 500, seed=9)
    chroms = ["chr%d" % (1 + k // 100) for k in range(pk.n_chunks)]
    b = ctx.upload(pk)
    b.run_nuc(10)
    b.run_occ()
    b.run_ins(0, 2000)
    for t in (L.T_OCC, L.T_NORM, L.T_SMOOTH, L.T_INS, L.T_OCC_PREFILL):
        text, ti = b.format_track(t, chroms, pk.chunk_start, compress=False)
        z, zi = b.format_track(t, chroms, pk.chunk_start, compress=True)
        text, z = text.tobytes(), z.tobytes()
        assert zi["text_bytes"] == len(text) and zi["lines"] == ti["lines"]
        assert gzip.GzipFile(fileobj=io.BytesIO(z + BGZF_EOF)).read() == text, t
        assert z == bgzf_lines_host(text), t                     # the kernels and the host restatement emit the same members
        print("track %d: %d lines, %.2f bytes of BGZF per line (text %.1f)" % (t, zi["lines"], len(z) / zi["lines"], len(text) / zi["lines"]))
    # a complete file: members + EOF marker, tabix index, region reads give the values back
    vals = b.track(L.T_OCC)
    z, _ = b.format_track(L.T_OCC, chroms, pk.chunk_start, compress=True)
    path = str(tmp_path / "occ.bedgraph.gz")
    with open(path, "wb") as fh:
        fh.write(z.tobytes() + BGZF_EOF)
    assert tabix_index(path) > 100000
    for k in (0, 150, 299):
        s = int(pk.chunk_start[k])
        tr = Track(chroms[k], s, s + 2120)
        tr.read_track(path)
        ref = vals[int(pk.out_off[k]):int(pk.out_off[k + 1])]
        m = ~np.isnan(ref)
        assert np.array_equal(np.isnan(tr.vals), ~m)
        assert np.array_equal(tr.vals[m], np.array([float(f2s(float(x))) for x in ref[m]]))
    b.free()


def test_device_writer_at_scale(ctx):
    """20,000 configs[2] chunks (42.4 Mbp): one call per track; text size = what the native writer produces, members inflate"""
    import time
    pk = make_synthetic_chunks(20000, 2120, 500, seed=3)
    chroms = ["chr%d" % (1 + k // 5000) for k in range(pk.n_chunks)]
    b = ctx.upload(pk)
    b.run_occ()
    t0 = time.perf_counter()
    z, info = b.format_track(L.T_OCC, chroms, pk.chunk_start, compress=True)
    dt = time.perf_counter() - t0
    print("device writer: %.1f Mbp track -> %d lines, %.1f MB of text, %.1f MB of BGZF in %.3f s (%.0f Mbp/s)" % (
        pk.total_bp / 1e6, info["lines"], info["text_bytes"] / 1e6, info["bytes"] / 1e6, dt, pk.total_bp / dt / 1e6))
    raw = gzip.GzipFile(fileobj=io.BytesIO(z.tobytes() + BGZF_EOF)).read()
    assert len(raw) == info["text_bytes"] and raw.count(b"\n") == info["lines"]
    vals = b.track(L.T_OCC)
    k = 12345
    seg = vals[int(pk.out_off[k]):int(pk.out_off[k]) + 3]
    first = ("%s\t%d\t%d\t%s\n" % (chroms[k], int(pk.chunk_start[k]), int(pk.chunk_start[k]) + 1, f2s(float(seg[0])))).encode()
    assert first in raw
    b.free()


def test_index_from_device_records_equals_index_from_the_file(ctx, tmp_path):
    """the .tbi built from the device's per-leaf-bin runs (natac_batch_format_index_*, writer.TbiBuilder) while the file is being
    assembled out of several results is byte-identical to the one natac_tabix_index builds by reading the finished file --
    float tracks (one record per base), integer tracks (long zero runs that span 16-kb windows: records in inner bins), several
    chromosomes, members that end inside records"""
    from nucleoatac_amd.writer import TbiBuilder
    subs = [make_synthetic_chunks(700, 2120, 500, seed=20 + i, first_chunk=700 * i) for i in range(3)]
    # sparse sub-batch: few fragments -> insertion track mostly zero, runs of equal values tens of kb long across chunk... (runs stop at chunk ends)
    sparse = make_synthetic_chunks(40, 60000, 30, seed=99, first_chunk=4000)
    subs.append(sparse)
    chrom_of = lambda i, pk: ["chr%d" % (1 + i)] * pk.n_chunks
    for track in (L.T_OCC, L.T_INS, L.T_SMOOTH):
        path = str(tmp_path / ("t%d.bedgraph.gz" % track))
        tb = TbiBuilder()
        off = 0
        with open(path, "wb") as fh:
            for i, pk in enumerate(subs):
                b = ctx.upload(pk)
                b.run_nuc(10)
                b.run_occ()
                b.run_ins(0, 2000)
                z, info = b.format_track(track, chrom_of(i, pk), pk.chunk_start, compress=True)
                fh.write(z.tobytes())
                tb.push(info["index"], off)
                off += len(z)
                assert len(info["index"]["cid"]) < info["lines"] / 50 + 2000     # runs, not lines
                b.free()
            fh.write(BGZF_EOF)
        n_dev = tb.write(path + ".dev.tbi")
        n_file = tabix_index(path)
        assert n_dev == n_file > 100000
        assert open(path + ".dev.tbi", "rb").read() == open(path + ".tbi", "rb").read(), track


def test_bias_scored_on_the_device_equals_host_packed_bias(ctx):
    """natac_batch_create_from_seq: the batch's Tn5 bias computed on the device from the sequence windows gives bit-identical
    tracks to the batch whose bias array was scored through natac_pwm_bias and uploaded (InsertionBiasTrack.computeBias,
    pyatac/bias.py:85-92)"""
    from helpers import synth_stores
    from nucleoatac_amd import set_context
    from nucleoatac_amd.pipeline import pack
    from nucleoatac_amd.pyatac.bias import PWM
    from nucleoatac_amd.pyatac.chunk import Chunk
    import nucleoatac_amd
    prev = nucleoatac_amd._default_ctx
    set_context(ctx)                       # pack() scores the host-side bias through the default context
    frags, fasta = synth_stores(11)
    chunks = [Chunk("chrS", s, s + 900 + 37 * i) for i, s in enumerate(range(1200, 11000, 1400))]
    pwm = PWM.open("Human")
    host = pack(chunks, frags, fasta, fasta.chrom_sizes(), pwm, window=121, upper=251)
    dev = pack(chunks, frags, fasta, fasta.chrom_sizes(), pwm, window=121, upper=251, bias_on_device=True)
    set_context(prev)
    assert dev.bias_log is None and dev.seq is not None and host.bias_log is not None
    outs = []
    for pk in (host, dev):
        b = ctx.upload(pk)
        b.run_nuc(10)
        b.run_occ()
        outs.append([b.track(t) for t in (L.T_BACKGROUND, L.T_NORM, L.T_OCC, L.T_OCC_LOWER)])
        b.free()
    for x, y in zip(*outs):
        assert np.array_equal(x, y, equal_nan=True)
    assert np.abs(outs[0][0]).max() > 0


def test_device_writer_on_random_tracks(ctx, tmp_path):
    """arbitrary values through the device writer (natac_batch_set_track): NaN runs of every length and position, runs of equal
    values (incl. +0 / -0, which compare equal and form one run), zeros, negative and tiny / huge magnitudes, integers, ragged
    chunk lengths, chromosome names of 1..64 characters, coordinates up to 2^40: text == native host writer for every
    write_zero / keep-runs-before-NaN mode, BGZF members inflate to it and equal the host restatement"""
    rng = np.random.default_rng(77)
    for rnd in range(6):
        nc = int(rng.integers(1, 40))
        lens = rng.integers(121, 3000, size=nc)
        nb = lens + 493
        pk = PackedChunks(np.sort(rng.integers(0, 2 ** 40 if rnd == 5 else 10 ** 9, size=nc)), lens, np.zeros(nc + 1, np.int64), np.zeros(0, np.int32),
                          np.zeros(0, np.int32), np.concatenate(([0], np.cumsum(nb))), np.zeros(int(nb.sum())))
        names = ["".join(rng.choice(list("abcXYZ_0123456789."), size=int(rng.integers(1, 65)))) for _ in range(4)]
        chroms = [names[int(i)] for i in np.sort(rng.integers(0, 4, size=nc))]
        n = int(lens.sum())
        kind = rng.integers(0, 6, size=n)
        v = np.where(kind == 0, rng.normal(0, 1, n), 0.0)
        v = np.where(kind == 1, rng.integers(-3, 50, n).astype(float), v)
        v = np.where(kind == 2, rng.normal(0, 1, n) * 10.0 ** rng.integers(-30, 11, n), v)
        v = np.where(kind == 3, np.round(rng.random(n), 2), v)
        v = np.where(kind == 4, -0.0, v)
        # runs: repeat values over random stretches; NaN stretches
        i = 0
        while i < n:
            ln = int(rng.integers(1, 60))
            mode = rng.integers(0, 5)
            if mode == 0:
                v[i:i + ln] = v[i]
            elif mode == 1:
                v[i:i + ln] = np.nan
            i += ln
        b = ctx.upload(pk)
        b.set_track(L.T_SMOOTH, v)
        for wz, keep in ((True, False), (False, False), (True, True), (False, True)):
            text, info = b.format_track(L.T_SMOOTH, chroms, pk.chunk_start, write_zero=wz, keep_runs_before_nan=keep, compress=False)
            want = _native_text(tmp_path, chroms, pk.chunk_start, pk.out_off, v, write_zero=wz, keep_runs_before_nan=keep)
            assert info["hard"] == 0 and text.tobytes() == want, (rnd, wz, keep)
        z, zi = b.format_track(L.T_SMOOTH, chroms, pk.chunk_start, compress=True)
        text, _ = b.format_track(L.T_SMOOTH, chroms, pk.chunk_start, compress=False)
        assert gzip.GzipFile(fileobj=io.BytesIO(z.tobytes() + BGZF_EOF)).read() == text.tobytes()
        assert z.tobytes() == bgzf_lines_host(text.tobytes())
        b.free()
    # an all-NaN track: nothing to write
    pk = make_synthetic_chunks(3, 500, 0, seed=1, counts=np.zeros(3, np.int64))
    b = ctx.upload(pk)
    b.set_track(L.T_OCC, np.full(pk.total_bp, np.nan))
    for comp in (False, True):
        out, info = b.format_track(L.T_OCC, ["c"] * 3, pk.chunk_start, compress=comp)
        assert len(out) == 0 and info["lines"] == 0
    b.free()


def test_results_fetched_while_the_next_track_is_formatted(ctx):
    """round 6: natac_batch_format_fetch_begin / _wait -- the copy of a finished result into pinned memory runs on a second stream while
    the batch formats its next track (what executor.PipelinedExecutor does with its five text tracks).  Same bytes and the same tabix
    records as the blocking fetch, for every track, in any interleaving; a batch freed with copies still in flight waits for them."""
    from nucleoatac_amd.device import pinned_empty
    pk = make_synthetic_chunks(600, 2120, 500, seed=17)
    chroms = ["chr%d" % (1 + k // 200) for k in range(pk.n_chunks)]
    b = ctx.upload(pk)
    b.run_nuc(10)
    b.run_occ()
    tracks = (L.T_NORM, L.T_SMOOTH, L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER)
    ref = {}
    for t in tracks:
        z, info = b.format_track(t, chroms, pk.chunk_start, compress=True)
        ref[t] = (z.copy(), info)
    slots = {t: pinned_empty(len(ref[t][0]) + 4096, np.uint8) for t in tracks}
    for s in slots.values():
        s[:] = 0xee
    got = {}
    for t in tracks:                                     # five copies begun, none waited for
        z, info = b.format_track(t, chroms, pk.chunk_start, compress=True, out=lambda n, t=t: slots[t][:n], wait=False)
        got[t] = (z, info)
    b.format_wait()
    for t in tracks:
        z, info = got[t]
        assert np.array_equal(z, ref[t][0]), t
        assert info["bytes"] == ref[t][1]["bytes"] and info["lines"] == ref[t][1]["lines"]
        for k in ("cid", "beg", "end", "count", "t0", "t1", "member_pos"):
            assert np.array_equal(info["index"][k], ref[t][1]["index"][k]), (t, k)
        assert (slots[t][len(z):] == 0xee).all()          # nothing written past the result
    b.format_wait()                                      # nothing pending: a no-op
    z, info = b.format_track(L.T_NORM, chroms, pk.chunk_start, compress=True, out=lambda n: slots[L.T_NORM][:n], wait=False)
    b.free()                                             # waits for the copy it still owns
    assert np.array_equal(z, ref[L.T_NORM][0])
