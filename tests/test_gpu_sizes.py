"""a3: insert-size histogram (getFragmentSizesFromChunkList / getAllFragmentSizes, pyatac/fragments.pyx:101-145;
FragmentSizes.calculateSizes, pyatac/fragmentsizes.py:22-27) on the GPU against reference-generated goldens
(tests/golden/sizes_hist.npz: overlapping, nested, adjacent, empty, chromosome-start and unsorted chunk lists), and
a4: the PWM bias kernel asserted directly against the reference's computeBias values."""
import numpy as np
import pytest

from helpers import golden, synth_stores

pytestmark = pytest.mark.gpu

SETS = ("disjoint", "overlap", "adjacent", "empty", "chromstart", "unsorted")
PARAMS = ((0, 251, 1), (30, 251, 1), (0, 2000, 1), (105, 251, 0))


@pytest.fixture(scope="module")
def ctx():
    from nucleoatac_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def test_fragment_sizes_bit_exact(ctx):
    g = golden("sizes_hist")
    for lo, up, atac in PARAMS:
        tag = "%d_%d_%d" % (lo, up, atac)
        l, n = (g["l"], g["n"]) if atac else (g["l"] - 4, g["n"] + 8)
        big = 1 << 40
        got = ctx.fragment_sizes(l, n, [-big], [big], lo, up)
        assert got.dtype == np.float64 and np.array_equal(got, g["all_" + tag]), tag
        for name in SETS:
            iv = g[name + "_chunks"]
            got = ctx.fragment_sizes(l, n, iv[:, 0], iv[:, 1], lo, up)
            assert np.array_equal(got, g["%s_%s" % (name, tag)]), (name, tag)
    # shuffled fragment order (the C-ABI does not require position-sorted input), no fragments, no chunks
    rng = np.random.default_rng(0)
    o = rng.permutation(len(g["l"]))
    iv = g["overlap_chunks"]
    assert np.array_equal(ctx.fragment_sizes(g["l"][o], g["n"][o], iv[:, 0], iv[:, 1], 0, 251), g["overlap_0_251_1"])
    assert not ctx.fragment_sizes(np.zeros(0, np.int64), np.zeros(0, np.int32), iv[:, 0], iv[:, 1], 0, 251).any()
    assert not ctx.fragment_sizes(g["l"], g["n"], [], [], 0, 251).any()
    # inverted interval: contains nothing (the reference's `center >= start and center < end`)
    assert not ctx.fragment_sizes(g["l"], g["n"], [5000], [4000], 0, 251).any()


def test_fragment_sizes_host_api():
    """the reference-named functions on a FragmentStore: same counts and the same normalised FragmentSizes.vals"""
    from nucleoatac_amd.pyatac.chunk import Chunk, ChunkList
    from nucleoatac_amd.pyatac.fragments import getAllFragmentSizes, getFragmentSizesFromChunkList
    from nucleoatac_amd.pyatac.fragmentsizes import FragmentSizes
    g = golden("sizes_hist")
    frags, _ = synth_stores(21, holes=[(9000, 9600)])
    for lo, up, atac in PARAMS:
        tag = "%d_%d_%d" % (lo, up, atac)
        assert np.array_equal(getAllFragmentSizes(frags, lo, up, atac=atac), g["all_" + tag])
        for name in SETS:
            chunks = ChunkList(*[Chunk("chrS", int(s), int(e)) for s, e in g[name + "_chunks"]])
            assert np.array_equal(getFragmentSizesFromChunkList(chunks, frags, lo, up, atac=atac), g["%s_%s" % (name, tag)])
            fs = FragmentSizes(lo, up, atac=bool(atac))
            fs.calculateSizes(frags, chunks=chunks)
            assert np.array_equal(fs.vals, g["%s_%s_norm" % (name, tag)]), (name, tag)


def test_fragment_sizes_many_chunks_matches_bruteforce(ctx):
    """segments that span more chunk bounds than the LDS stage (global-memory search path) + heavily overlapping chunks"""
    rng = np.random.default_rng(5)
    nf, nchunks = 200000, 60000
    l = np.sort(rng.integers(0, 3000000, nf)).astype(np.int64)
    n = rng.integers(-5, 400, nf).astype(np.int32)
    cs = rng.integers(0, 3000000, nchunks).astype(np.int64)
    ce = cs + rng.integers(0, 5000, nchunks)
    got = ctx.fragment_sizes(l, n, cs, ce, 0, 251)
    c = l + (n.astype(np.int64) - 1) // 2
    ok = (n >= 0) & (n < 251)
    cnt = np.searchsorted(np.sort(cs), c, "right") - np.searchsorted(np.sort(ce), c, "right")
    ref = np.bincount(n[ok], weights=cnt[ok].astype(np.float64), minlength=251)[:251]
    assert np.array_equal(got, ref)
    # literal per-chunk loop (the reference's order of evaluation) on a subsample of the chunks
    sub = slice(0, 300)
    ref2 = np.zeros(251)
    for s, e in zip(cs[sub], ce[sub]):
        m = ok & (c >= s) & (c < e)
        np.add.at(ref2, n[m], 1.0)
    assert np.array_equal(ctx.fragment_sizes(l, n, cs[sub], ce[sub], 0, 251), ref2)


def test_fragment_sizes_config3_scale_under_5ms(ctx):
    """BASELINE configs[2] scale: 50 M fragments x 100 k chunks; kernel time from the library's HIP events"""
    rng = np.random.default_rng(1)
    nchunks, L = 100000, 2120
    cs = np.arange(nchunks, dtype=np.int64) * 2300 + 1000
    ce = cs + L
    nf = 50_000_000
    which = np.sort(rng.integers(0, nchunks, nf))
    l = (cs[which] + rng.integers(-126, L + 126, nf)).astype(np.int64)
    l.sort()
    n = rng.integers(20, 700, nf).astype(np.int32)
    ctx.profile_enable(True)
    ctx.profile_reset()
    got = ctx.fragment_sizes(l, n, cs, ce, 0, 251)
    ms, launches = ctx.profile()["size_hist"]
    ctx.profile_enable(False)
    c = l + (n.astype(np.int64) - 1) // 2
    ok = (n >= 0) & (n < 251)
    k = np.searchsorted(cs, c[ok], "right") - 1
    inside = (k >= 0) & (c[ok] < ce[np.maximum(k, 0)])
    ref = np.bincount(n[ok][inside], minlength=251)[:251].astype(np.float64)
    assert np.array_equal(got, ref)
    assert launches == 1 and ms < 5.0, "size histogram kernel took %.2f ms" % ms
    print("size_hist: %.3f ms for %d fragments x %d chunks" % (ms, nf, nchunks))


@pytest.mark.parametrize("case", ["chunks_basic", "chunks_gaps"])
def test_pwm_bias_matches_reference_directly(ctx, case):
    """a4: natac_pwm_score against the reference's InsertionBiasTrack.computeBias values (bias.py:85-92), rtol 1e-12"""
    p = golden("params_example")
    g = golden(case)
    for k in range(int(g["n_chunks"])):
        seq = bytes(g["c%d_seq" % k]).decode()
        got = ctx.pwm_bias(seq, p["pwm_mat"], [str(x) for x in p["pwm_nucleotides"]])
        ref = g["c%d_bias_log" % k]
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-13)


def test_cli_occ_and_nuc_without_sizes(tmp_path):
    """`nucleoatac occ` / `nuc` without --sizes compute the size distribution from the BAM over the slopped + merged
    chunk list (run_occ.py:95, run_nuc.py:158-159): the saved fragmentsizes.txt equals the oracle's histogram"""
    from nucleoatac_amd.nucleoatac.cli import main
    from nucleoatac_amd.pyatac.fragmentsizes import FragmentSizes
    from oracle import natac_oracle as O
    g = golden("sizes_hist")
    par = golden("params_example")
    frags, fasta = synth_stores(21, holes=[(9000, 9600)])
    bam = str(tmp_path / "synth.npz")
    frags.save_npz(bam)
    fa = str(tmp_path / "synth.fa")
    with open(fa, "w") as f:
        f.write(">chrS\n")
        s = fasta.seqs["chrS"].tobytes().decode()
        for i in range(0, len(s), 60):
            f.write(s[i:i + 60] + "\n")
    regions = [(1060, 1743), (1700, 2500), (4860, 5440), (8060, 8840)]        # the first two merge after the +-60 slop
    bed = str(tmp_path / "r.bed")
    with open(bed, "w") as f:
        for s_, e_ in regions:
            f.write("chrS\t%d\t%d\n" % (s_, e_))
    merged = [(1000, 2560), (4800, 5500), (8000, 8900)]
    ref = O.normalise_sizes(O.fragment_sizes_from_chunks(g["l"], g["n"], [a for a, _ in merged], [b for _, b in merged], 0, 251))
    out = str(tmp_path / "t")
    main(["occ", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out])
    fs = FragmentSizes.open(out + ".fragmentsizes.txt")
    assert fs.lower == 0 and fs.upper == 251
    np.testing.assert_allclose(fs.get(), ref, rtol=1e-11, atol=0)            # 12 significant digits in the text file
    vm = str(tmp_path / "v.npz")
    np.savez(vm, vmat=par["vmat"], vlower=par["vlower"], vupper=par["vupper"])
    out2 = str(tmp_path / "u")
    main(["nuc", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out2, "--vmat", vm,
          "--occ_track", out + ".occ.bedgraph.gz"])
    import os
    assert os.path.exists(out2 + ".nucpos.bed.gz") and os.path.exists(out2 + ".nucleoatac_signal.bedgraph.gz")
