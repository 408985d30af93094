"""CPU: packed chunk layout, synthetic workloads, sharding helpers."""
import numpy as np
import pytest

from nucleoatac_amd.packing import PackedChunks, pack_chunks, sort_by_centre
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution


def test_synthetic_is_seeded_and_sorted():
    a = make_synthetic_chunks(50, 2120, 500, seed=3)
    b = make_synthetic_chunks(50, 2120, 500, seed=3)
    assert np.array_equal(a.frag_lpos, b.frag_lpos) and np.array_equal(a.bias_log, b.bias_log)
    assert a.n_chunks == 50 and a.n_frags == 25000 and a.total_bp == 50 * 2120
    for k in range(a.n_chunks):
        l, n = a.chunk_frags(k)
        c = l.astype(np.int64) + (n.astype(np.int64) - 1) // 2
        assert np.all(np.diff(c) >= 0)
        assert c.min() >= -126 and c.max() < 2120 + 126
        assert len(a.chunk_bias(k)) == 2120 + 246 + 247
    p = make_synthetic_chunks(200, 700, 100, seed=1, poisson=True)
    assert p.n_frags == int(np.diff(p.frag_off).sum()) and np.diff(p.frag_off).std() > 0


def test_subset_is_consistent():
    a = make_synthetic_chunks(20, 600, 100, seed=4)
    s = a.subset(5, 12)
    assert s.n_chunks == 7 and s.total_bp == 7 * 600
    for k in range(7):
        assert np.array_equal(s.chunk_frags(k)[0], a.chunk_frags(5 + k)[0])
        assert np.array_equal(s.chunk_bias(k), a.chunk_bias(5 + k))


def test_pack_chunks_from_sorted_fragments():
    rng = np.random.default_rng(0)
    l = np.sort(rng.integers(0, 20000, size=3000))
    n = rng.integers(20, 400, size=3000)
    bias = rng.normal(size=21000)
    pk = pack_chunks([("chr1", 2000, 3000), ("chr1", 7000, 9000)], {"chr1": l}, {"chr1": n},
                     bias_tracks={"chr1": (0, bias)})
    assert pk.n_chunks == 2 and list(pk.chunk_len) == [1000, 2000]
    for k, (s, e) in enumerate(((2000, 3000), (7000, 9000))):
        lr, nr = pk.chunk_frags(k)
        assert np.all(lr + s >= s - 2126) and np.all(lr + s < e + 2126)
        o = sort_by_centre(lr, nr)
        assert np.array_equal(o, np.arange(len(o)))
        assert np.array_equal(pk.chunk_bias(k), bias[s - 246:e + 247])
    with pytest.raises(ValueError):
        pack_chunks([("chr1", 100, 300)], {"chr1": l}, {"chr1": n}, bias_tracks={"chr1": (0, bias)})


def test_validation_errors():
    with pytest.raises(ValueError):
        PackedChunks(chunk_start=[0], chunk_len=[10], frag_off=[0, 2], frag_lpos=[1], frag_ilen=[1], bias_off=None, bias_log=None)
    with pytest.raises(ValueError):
        PackedChunks(chunk_start=[0], chunk_len=[0], frag_off=[0, 0], frag_lpos=[], frag_ilen=[], bias_off=None, bias_log=None)
    with pytest.raises(ValueError):
        PackedChunks(chunk_start=[0], chunk_len=[10], frag_off=[0, 0], frag_lpos=[], frag_ilen=[], bias_off=[0, 5],
                     bias_log=np.zeros(5))


def test_synthetic_distributions():
    s = synth_size_distribution(251)
    a, b = synth_occ_distributions(251)
    assert abs(s.sum() - 1) < 1e-12 and abs(a.sum() - 1) < 1e-12 and abs(b.sum() - 1) < 1e-12
    assert (a > 0).all() and (b > 0).all()


def test_native_pack_matches_per_chunk_fetch():
    """pipeline.pack (natac_pack_chunks) == the reference-shaped per-chunk fetch + shift + stable centre sort"""
    from nucleoatac_amd import pipeline
    from nucleoatac_amd.packing import sort_by_centre
    from nucleoatac_amd.pyatac.chunk import Chunk
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    rng = np.random.default_rng(0)
    sizes = {"chrA": 600_000, "chrB": 200_000, "chrE": 5000}
    pos = {c: np.sort(rng.integers(0, sizes[c], n)) for c, n in (("chrA", 60000), ("chrB", 10000), ("chrE", 0))}
    tl = {c: rng.integers(1, 700, len(pos[c])) * rng.choice([-1, 1], len(pos[c])) for c in pos}
    st = FragmentStore(list(sizes), list(sizes.values()), pos, tl)
    chunks = [Chunk("chrA", int(s), int(s) + int(rng.integers(200, 3000))) for s in np.sort(rng.integers(3000, 590_000, 300))]
    chunks += [Chunk("chrB", int(s), int(s) + 2120) for s in np.sort(rng.integers(3000, 190_000, 100))]
    chunks += [Chunk("chrE", 100, 900), Chunk("chrZ", 100, 900), Chunk("chrA", 0, 300)]
    for atac in (True, False):
        pk = pipeline.pack(chunks, st, atac=atac)
        offs, ls, ns = [0], [], []
        for ch in chunks:
            l, n = st.fetch(ch.chrom, ch.start - pipeline.MARGIN, ch.end + pipeline.MARGIN, 1 if atac else 0)
            keep = l >= ch.start - pipeline.MARGIN
            l, n = l[keep], n[keep]
            lr = (l - ch.start).astype(np.int32)
            o = sort_by_centre(lr, n)
            ls.append(lr[o])
            ns.append(n[o].astype(np.int32))
            offs.append(offs[-1] + len(lr))
        assert np.array_equal(pk.frag_off, np.array(offs))
        assert np.array_equal(pk.frag_lpos, np.concatenate(ls)) and np.array_equal(pk.frag_ilen, np.concatenate(ns))
        assert pk.chroms == [c.chrom for c in chunks]
        pk.validate()
    assert pipeline.pack([], st).n_chunks == 0


def test_sequence_windows_pack_and_host_helpers():
    """pack(bias_on_device=True): sequence windows instead of a scored bias array (natac_batch_create_from_seq), their validation and
    subsets; chunk_fragment_counts == the per-chunk fetch it replaced; prefetch_map keeps order and forwards exceptions"""
    from helpers import synth_stores
    from nucleoatac_amd.pipeline import chunk_fragment_counts, pack, prefetch_map
    from nucleoatac_amd.pyatac.bias import PWM
    from nucleoatac_amd.pyatac.chunk import Chunk
    frags, fasta = synth_stores(11)
    chunks = [Chunk("chrS", s, s + 700 + 13 * i) for i, s in enumerate(range(1200, 11000, 1400))]
    pwm = PWM.open("Human")
    K = pwm.up + pwm.down + 1
    pk = pack(chunks, frags, fasta, fasta.chrom_sizes(), pwm, window=121, upper=251, bias_on_device=True)
    assert pk.bias_log is None and pk.bias_off is None and pk.seq.dtype == np.uint8 and pk.pwm_log.shape == (4, K)
    for k, ch in enumerate(chunks):
        a, b = int(pk.seq_off[k]), int(pk.seq_off[k + 1])
        assert b - a == ch.length() + 246 + 247 + K - 1
        assert pk.seq[a:b].tobytes() == fasta.seqs["chrS"][ch.start - 246 - pwm.up:ch.end + 247 + pwm.down].tobytes()
    sub = pk.subset(2, 5)
    assert sub.n_chunks == 3 and sub.seq[:10].tobytes() == pk.seq[int(pk.seq_off[2]):int(pk.seq_off[2]) + 10].tobytes()
    with pytest.raises(ValueError):
        PackedChunks(pk.chunk_start, pk.chunk_len, pk.frag_off, pk.frag_lpos, pk.frag_ilen, None, None, seq_off=pk.seq_off,
                     seq=pk.seq[:-1], pwm_log=pk.pwm_log, pwm_nucs=pk.pwm_nucs)
    with pytest.raises(Exception) as e:
        pack([Chunk("chrS", 100, 900)], frags, fasta, fasta.chrom_sizes(), pwm, window=121, upper=251, bias_on_device=True)
    assert "too close to the chromosome end" in str(e.value)
    # shard-balance counts: one searchsorted per chromosome == FragmentStore.fetch per chunk
    want = [len(frags.fetch(c.chrom, c.start, c.end)[0]) for c in chunks] + [0]
    assert list(chunk_fragment_counts(frags, chunks + [Chunk("chrNone", 5, 50)])) == want
    # prefetch_map: order, laziness bounded by depth, exceptions
    seen = []
    def f(x):
        seen.append(x)
        if x == 7:
            raise KeyError("boom")
        return x * x
    it = prefetch_map(f, range(20), depth=3)
    assert [next(it) for _ in range(5)] == [0, 1, 4, 9, 16] and len(seen) <= 8
    with pytest.raises(KeyError):
        list(it)
    assert list(prefetch_map(lambda x: x + 1, [], depth=2)) == []
