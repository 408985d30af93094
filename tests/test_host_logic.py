"""CPU: host-side mirror of the reference API (no GPU needed): intervals, text formats, peak logic, parsers."""
import io
import os

import numpy as np
import pytest

from helpers import GOLDEN, golden
from nucleoatac_amd.pyatac.bias import PWM
from nucleoatac_amd.pyatac.chunk import Chunk, ChunkList
from nucleoatac_amd.pyatac.chunkmat2d import ChunkMat2D, FragmentMat2D
from nucleoatac_amd.pyatac.fragments import FragmentStore
from nucleoatac_amd.pyatac.fragmentsizes import FragmentSizes
from nucleoatac_amd.pyatac.tracks import Track
from nucleoatac_amd.pyatac.utils import call_peaks, reduce_peaks
from nucleoatac_amd.pyatac.VMat import VMat, VMat_Error

BED = os.path.join(GOLDEN, "ref_example.bed")
CHRS = {"chrII": 813184, "chrIV": 1531933, "chrVII": 1090940, "chrX": 745751, "chrXII": 1078177, "chrXV": 1091291,
        "chrXVI": 948066, "chrI": 230218, "chrIII": 316620, "chrV": 576874, "chrVI": 270161, "chrVIII": 562643,
        "chrIX": 439888, "chrXI": 666816, "chrXIII": 924431, "chrXIV": 784333, "chrM": 85779}


def test_call_peaks_reference_cases():
    """the reference's tests/test_utils.py:8-19"""
    sig = np.array([1, 2, 3, 2, 1, 4, 1, 2, 1, 0, 0])
    assert np.array_equal(call_peaks(sig.copy(), min_signal=1, sep=3), [2, 5])
    assert np.array_equal(call_peaks(sig.copy(), min_signal=1, sep=1), [2, 5, 7])
    assert np.array_equal(call_peaks(sig.copy(), min_signal=3, sep=2), [2, 5])


def test_call_peaks_fills_nan_in_place():
    x = np.array([np.nan, 1.0, 3.0, 1.0, np.nan, 0.5, 2.0, 0.5, np.nan])
    call_peaks(x, sep=2, boundary=0)
    assert not np.isnan(x).any() and x[0] == 0.5
    assert call_peaks(np.full(5, np.nan)).size == 0
    assert np.array_equal(reduce_peaks(np.array([10, 20, 25, 60]), [1.0, 5.0, 4.0, 2.0], 30), [20, 60])


def test_chunklist_read_slop_merge_split():
    chunks = ChunkList.read(BED, chromDict=CHRS, min_offset=255)
    assert len(chunks) == 19 and chunks[0].chrom == "chrII" and (chunks[0].start, chunks[0].end) == (706612, 707705)
    n0 = chunks[0].length()
    chunks.slop(CHRS, up=60, down=60)
    assert chunks[0].length() == n0 + 120
    chunks.merge()
    assert chunks.isSorted()
    sets = chunks.split(items=5)
    assert sum(len(s) for s in sets) == len(chunks) and len(sets[0]) == 5
    assert sum(len(s) for s in chunks.split(bases=5000)) == len(chunks)
    with pytest.raises(Exception):
        chunks.split()
    # merge joins regions closer than sep (+1): gap of exactly `sep` merges
    cl = ChunkList(Chunk("c", 0, 100), Chunk("c", 100, 150), Chunk("c", 152, 160), Chunk("d", 0, 5))
    cl.merge(sep=0)
    assert [(c.chrom, c.start, c.end) for c in cl] == [("c", 0, 150), ("c", 152, 160), ("d", 0, 5)]
    with pytest.raises(ValueError):
        cl.append("not a chunk")
    # min_offset clipping / min_length filter / unknown chromosomes
    with pytest.warns(UserWarning):
        few = ChunkList.read(BED, chromDict={"chrII": 813184}, min_offset=255, min_length=1000)
    assert all(c.chrom == "chrII" and c.length() >= 1000 for c in few)


def test_chunk_slop_strand_and_center():
    c = Chunk("chr1", 100, 200, strand="-")
    c.slop({"chr1": 1000}, up=10, down=30)
    assert (c.start, c.end) == (70, 210)
    d = Chunk("chr1", 5, 20).slop({"chr1": 22}, up=10, down=10, new=True)
    assert (d.start, d.end) == (0, 22)
    e = Chunk("chr1", 10, 21)
    e.center()
    assert (e.start, e.end) == (15, 16)
    assert Chunk("chr1", 1, 2, name="n").asBed() == "chr1\t1\t2\t1\tn\t*"


def test_track_write_bedgraph_format():
    """run-length text with python-2 float formatting, NaN runs skipped, and the reference's rule that a run directly
    followed by a NaN is never flushed (pyatac/tracks.py:56-66; rows pinned by tests/golden/write_track_rows.npz)"""
    t = Track("chr1", 10, 20, vals=np.array([0, 0, 1.5, 1.5, np.nan, np.nan, 1 / 3.0, 1 / 3.0, 0, 0]))
    h = io.StringIO()
    t.write_track(h)
    assert h.getvalue() == "chr1\t10\t12\t0.0\nchr1\t16\t18\t0.333333333333\nchr1\t18\t20\t0.0\n"
    h = io.StringIO()
    t.write_track(h, write_zero=False)
    assert h.getvalue() == "chr1\t16\t18\t0.333333333333\n"
    h = io.StringIO()
    t.write_track(h, keep_runs_before_nan=True)
    assert h.getvalue() == "chr1\t10\t12\t0.0\nchr1\t12\t14\t1.5\nchr1\t16\t18\t0.333333333333\nchr1\t18\t20\t0.0\n"
    with pytest.raises(Exception):
        t.write_track(io.StringIO(), vals=np.zeros(3))
    with pytest.raises(Exception):
        Track("chr1", 0, 5, vals=[1, 2])
    assert t.get(pos=12) == 1.5 and len(t.get(12, 16)) == 4


def test_track_read_example_scores():
    """the reference's tests/test_tracks.py:31-36 pin"""
    chunk = ChunkList.read(BED)[0]
    t = Track(chunk.chrom, chunk.start, chunk.end)
    t.read_track(os.path.join(GOLDEN, "ref_example.Scores.bedgraph.gz"))
    assert abs(1.35994655714 - t.get(pos=706661)) < 0.001


def test_bam_decoder_on_reference_fixture():
    st = FragmentStore.from_bam(os.path.join(GOLDEN, "ref_single_read.bam"))
    sr = golden("single_read")
    l, n = st.fetch("chrII", int(sr["start"]), int(sr["end"]))
    assert np.array_equal(l, sr["l"]) and np.array_equal(n, sr["n"])
    assert st.chrom_sizes()["chrII"] == 813184 and len(st.references) == 17
    with pytest.raises(ValueError):
        FragmentStore.open("reads.sam")


def test_vmat_text_roundtrip(tmp_path):
    v = VMat.open(os.path.join(GOLDEN, "ref_example.VMat"))
    assert v.mat.shape == (130, 121) and (v.lower, v.upper, v.w) == (115, 245, 60)
    d = VMat.default()
    assert d.mat.shape == (146, 121) and (d.lower, d.upper, d.w) == (105, 251, 60)
    p = str(tmp_path / "x.VMat")
    d.save(p)
    r = VMat.open(p)
    assert np.allclose(r.mat, d.mat, rtol=1e-11) and r.lower == 105
    d.trim(110, 250, 50)
    assert d.mat.shape == (140, 101) and d.w == 50
    with pytest.raises(VMat_Error):
        d.trim(100, 250, 50)
    with pytest.raises(VMat_Error):
        VMat(np.zeros((3, 5)), 0, 4)


def test_pwm_and_fragmentsizes_io(tmp_path):
    p = PWM.open("Human")
    assert p.mat.shape == (4, 21) and (p.up, p.down) == (10, 10) and p.nucleotides == ["A", "C", "G", "T"]
    f = str(tmp_path / "h.PWM.txt")
    p.save(f)
    q = PWM.open(f)
    assert np.allclose(q.mat, p.mat) and q.nucleotides == p.nucleotides
    fs = FragmentSizes(0, 5, vals=np.array([0.0, 0.25, 0.5, 0.125, 0.125]))
    g = str(tmp_path / "s.txt")
    fs.save(g)
    back = FragmentSizes.open(g)
    assert np.array_equal(back.get(), fs.get()) and back.get(size=2) == 0.5 and list(back.get(1, 3)) == [0.25, 0.5]


def test_chunkmat2d_get_and_getins():
    """the reference's tests/test_chunkmat2d.py:13-18 + getIns geometry"""
    x = FragmentMat2D("chr1", 500, 1000, 0, 200)
    x.mat[100, 5] = 1
    assert np.array_equal(x.get(start=505, end=507, lower=100, upper=102), np.array([[1, 0], [0, 0]]))
    g = golden("ins_edge")
    m = ChunkMat2D("chrS", int(g["mat_start"]), int(g["mat_end"]), 0, 251)
    m.mat[g["mat_rows"], g["mat_cols"]] = g["mat_vals"]
    ins = m.getIns()
    assert np.array_equal(ins.vals, g["getins"]) and ins.start == int(g["getins_start"])


def test_cli_defaults_match_reference():
    from nucleoatac_amd.nucleoatac.cli import nucleoatac_parser
    a = nucleoatac_parser().parse_args(["occ", "--bed", "b", "--bam", "x", "--out", "o"])
    assert (a.upper, a.flank, a.min_occ, a.nuc_sep, a.confidence_interval, a.step, a.pwm) == (251, 60, 0.1, 120, 0.9, 5, "Human")
    n = nucleoatac_parser().parse_args(["nuc", "--bed", "b", "--bam", "x", "--out", "o", "--vmat", "v"])
    assert (n.min_z, n.min_lr, n.nuc_sep, n.redundant_sep, n.sd, n.atac, n.write_all) == (3, 0, 120, 25, 10, True, False)


def test_batched_finite_differences_reproduce_scipys_gradient():
    """fit_fuzz_one hands L-BFGS-B a gradient that is scipy's own 2-point finite difference, evaluated in one batch: the
    optimiser must take exactly the same path (bit-identical fuzz / weight / position, also next to a bound)"""
    from nucleoatac_amd.nucleoatac import NucleosomeCalling as N
    rng = np.random.default_rng(0)
    tasks = []
    for i in range(6):
        Lc = 1500
        x = np.arange(Lc)
        keys = np.sort(rng.choice(np.arange(100, Lc - 100, 35), size=14, replace=False))
        sd = rng.uniform(2.5, 45, size=len(keys))            # some fits end on the variance bounds (2^2, 50^2)
        v = sum(rng.uniform(0.5, 2) * np.exp(-0.5 * ((x - k) / s) ** 2) for k, s in zip(keys, sd)) + rng.normal(0, 0.01, Lc)
        tasks.append((v, keys, 120, 10))
    res = {}
    try:
        N.LOCKSTEP = False
        for fast in (False, True):
            N.FAST_FD = fast
            res[fast] = np.array([r for tk in tasks for r in N.fit_fuzz_chunk(tk)])
    finally:
        N.FAST_FD = N.LOCKSTEP = True
    assert res[True].shape == (6 * 14, 3)
    assert np.array_equal(res[False], res[True])


def test_lockstep_fits_take_the_optimisers_own_steps():
    """fuzzfit.fit_many advances many L-BFGS-B runs together through scipy's reverse-communication routine and evaluates
    their objectives in one set of numpy calls: every fit must end on the bits scipy.optimize.minimize gives one at a time
    (1-, 2- and 3-Gaussian fits, windows of every length, fits ending on a bound, flat and negative signal)"""
    from nucleoatac_amd.nucleoatac import NucleosomeCalling as N, fuzzfit
    assert fuzzfit.available(), "this scipy's private setulb no longer matches: the per-call path is used (slower, same values)"
    rng = np.random.default_rng(7)
    tasks = []
    for i in range(40):
        Lc = int(rng.integers(400, 2500))
        x = np.arange(Lc)
        nk = int(rng.integers(1, 18))
        keys = np.sort(rng.choice(np.arange(60, Lc - 60, 25), size=min(nk, (Lc - 120) // 25), replace=False))
        sd = rng.uniform(2.5, 45, size=len(keys))
        v = sum(rng.uniform(0.2, 3) * np.exp(-0.5 * ((x - k) / s) ** 2) for k, s in zip(keys, sd)) + rng.normal(0, 0.03, Lc)
        if i % 7 == 0:
            v -= 0.1                                   # parts of the window clamp at 0
        tasks.append((v, keys, 120, int(rng.choice([10, 25]))))
    try:
        N.LOCKSTEP = False
        one = [N.fit_fuzz_chunk(tk) for tk in tasks]
    finally:
        N.LOCKSTEP = True
    for group in (3, 64):
        old, fuzzfit.GROUP = fuzzfit.GROUP, group
        try:
            many = N.fit_fuzz_chunks(tasks)
        finally:
            fuzzfit.GROUP = old
        assert [len(m) for m in many] == [len(tk[1]) for tk in tasks]
        assert np.array_equal(np.array([r for c in one for r in c], dtype=np.float64),
                              np.array([r for c in many for r in c], dtype=np.float64))
    old, fuzzfit.MAXFUN = fuzzfit.MAXFUN, 25           # fits that come near scipy's evaluation limit are handed to scipy itself
    try:
        few = N.fit_fuzz_chunks(tasks[:6])
    finally:
        fuzzfit.MAXFUN = old
    assert np.array_equal(np.array([r for c in one[:6] for r in c], dtype=np.float64), np.array([r for c in few for r in c], dtype=np.float64))
    flat = (np.zeros(300), np.array([150]), 120, 10)   # an all-zero window has no valid weight bound: both paths refuse it
    for lock in (False, True):
        N.LOCKSTEP = lock
        try:
            with pytest.raises(ValueError, match="upper bound is less"):
                N.fit_fuzz_chunk(flat)
        finally:
            N.LOCKSTEP = True
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(2) as pool:                # the slicing over a --cores pool keeps the task order
        sliced = N.fit_fuzz_tasks(tasks, pool, 2)
    assert np.array_equal(np.array([r for c in many for r in c]), np.array([r for c in sliced for r in c]))


def test_native_fasta_loader_equals_the_line_loop(tmp_path):
    """natac_fasta_* (csrc/natac_fasta.hpp: header scan, counts and copies on threads) against the Python line loop of
    FastaStore.open: names up to the first blank, lower case, CRLF, empty records, no final newline, a record > 1 MiB"""
    import gzip
    from nucleoatac_amd.pyatac import seq as S
    rng = np.random.default_rng(3)
    letters = np.frombuffer(b"ACGTacgtNn", dtype=np.uint8)
    recs = [("chr1 some description", 2_500_000, 60, "\n"), ("chrEmpty", 0, 60, "\n"), ("chr2\tx", 70_001, 80, "\r\n"),
            ("scaffold_3", 59, 60, "\n"), ("chrLast", 1234, 50, "\n")]
    text, want = [], {}
    for name, n, width, eol in recs:
        s = letters[rng.integers(0, len(letters), n)].tobytes().decode()
        want[name.split()[0]] = s.upper()
        text.append(">" + name + eol + "".join(s[i:i + width] + eol for i in range(0, n, width)))
    body = "".join(text)
    body = body[:-1]                                       # no newline at the end of the file
    plain, gz = str(tmp_path / "g.fa"), str(tmp_path / "g.fa.gz")
    open(plain, "w", newline="").write(body)
    gzip.open(gz, "wt", newline="").write(body)
    assert S.FastaStore._native_ok()
    nat = S.FastaStore.open(plain)                         # native loader
    py = S.FastaStore.open(gz)                             # the line loop (gzip)
    assert nat.references == py.references == [r[0].split()[0] for r in recs]
    assert list(nat.lengths) == list(py.lengths) == [r[1] for r in recs]
    for c in nat.references:
        assert nat.seqs[c].tobytes().decode() == want[c] == py.seqs[c].tobytes().decode(), c
    assert nat.fetch("chr2", 100, 130) == want["chr2"][100:130]


def test_fasta_sizes_without_loading_and_prefetch(tmp_path):
    """FastaStore.sizes reads the record lengths from the .npy headers of an .npz / from a .fai, and falls back to the loaded
    store; FastaStore.prefetch + open give the store a plain open gives"""
    from nucleoatac_amd.pyatac import seq as S
    rng = np.random.default_rng(1)
    seqs = {"chrI": rng.integers(65, 70, 5000).astype(np.uint8), "chrII_b": rng.integers(65, 70, 123).astype(np.uint8)}
    npz = str(tmp_path / "g.npz")
    np.savez(npz, chrom_names=np.array(list(seqs)), **{"seq_" + c: a for c, a in seqs.items()})
    assert S.FastaStore.sizes(npz) == {"chrI": 5000, "chrII_b": 123}
    assert npz not in S._CACHE                              # nothing was loaded for that
    S.FastaStore.prefetch(npz)
    st = S.FastaStore.open(npz)
    assert st.chrom_sizes() == {"chrI": 5000, "chrII_b": 123} and np.array_equal(st.seqs["chrI"], seqs["chrI"])
    fa = str(tmp_path / "t.fa")
    open(fa, "w").write(">a x\nACGTAC\nGT\n>b\nAC\n")
    assert S.FastaStore.sizes(fa) == {"a": 8, "b": 2}       # no .fai: the loaded store answers
    fa2 = str(tmp_path / "u.fa")
    open(fa2, "w").write(">a x\nACGTAC\nGT\n>b\nAC\n")
    open(fa2 + ".fai", "w").write("a\t8\t5\t6\t7\nb\t2\t18\t2\t3\n")
    assert S.FastaStore.sizes(fa2) == {"a": 8, "b": 2} and fa2 not in S._CACHE


def test_sub_batches_cut_by_bases_and_by_count():
    """pipeline.sub_batches: consecutive slices that cover the list in order, each <= max_chunks chunks and -- unless a single chunk
    is longer -- <= target_bp bases (the rule `occ` cuts its pipeline's sub-batches by)"""
    from nucleoatac_amd.pipeline import SUB_BATCH_BP, sub_batches
    from nucleoatac_amd.pyatac.chunk import Chunk, ChunkList
    rng = np.random.default_rng(4)
    for lens, max_chunks, target in (([2120] * 10000, 4096, SUB_BATCH_BP), ([10120] * 3000, 4096, SUB_BATCH_BP),
                                     (list(rng.integers(121, 60000, 500)), 64, 200000), ([50_000_000, 100, 100], 4096, SUB_BATCH_BP), ([], 10, 10)):
        cl = ChunkList(*[Chunk("c", 100000 * i, 100000 * i + int(n)) for i, n in enumerate(lens)])
        parts = sub_batches(cl, max_chunks, target)
        flat = [c for p in parts for c in p]
        assert [c.start for c in flat] == [c.start for c in cl] and all(len(p) > 0 for p in parts)
        for p in parts:
            bp = sum(c.end - c.start for c in p)
            assert len(p) <= max_chunks and (bp <= target or len(p) == 1)
        # greedy: a slice could not have taken the next chunk as well
        for p, q in zip(parts[:-1], parts[1:]):
            assert len(p) == max_chunks or sum(c.end - c.start for c in p) + (q[0].end - q[0].start) > target
    assert len(sub_batches(ChunkList(*[Chunk("c", 0, 2120)] * 5000), 4096, SUB_BATCH_BP)) == 3       # 2,122 chunks of 2 kb per ~4.5 Mbp


def test_writer_two_phase_lookahead_order():
    """run_occ._Writer with a (start, finish) pair: start(k) runs before the result's buffers are released, finish(k) only after
    start(k + 1), every finish in result order, the last one at finish(); a failing start surfaces on the caller's thread"""
    from nucleoatac_amd.nucleoatac.run_occ import _Writer
    log = []

    class R(object):
        def __init__(self, seq):
            self.seq, self.tag, self.text, self.tracks, self.released = seq, [], {}, {}, False

        def release(self):
            if not self.released:
                log.append(("release", self.seq))
            self.released = True

    def start(r):
        assert not r.released
        log.append(("start", r.seq))
        return r.seq

    def finish(k):
        log.append(("finish", k))

    w = _Writer({}, {}, (start, finish), 4, True)
    w.start()
    for k in range(4):
        w.put(R(k))
    w.finish()
    pos = {e: i for i, e in enumerate(log)}
    for k in range(4):
        assert pos[("start", k)] < pos[("release", k)] < pos[("finish", k)]
        if k < 3:
            assert pos[("start", k + 1)] < pos[("finish", k)] < pos[("finish", k + 1)]
    assert log[-1] == ("finish", 3)

    def bad(r):
        raise ValueError("boom %d" % r.seq)

    w = _Writer({}, {}, (bad, finish), 2, True)
    w.start()
    w.put(R(0))
    with pytest.raises(ValueError, match="boom 0"):
        w.put(R(1))
        w.finish()


def test_occstore_interval_index():
    """occstore.OccTrackStore.locate: a region is served only when it lies inside ONE stored chunk; offsets follow the batch layout"""
    from nucleoatac_amd import occstore
    from nucleoatac_amd.pyatac.chunk import Chunk
    st = occstore.OccTrackStore()
    try:
        part0 = [Chunk("chr1", 1000, 3120), Chunk("chr1", 5000, 7120), Chunk("chr2", 100, 900)]
        part1 = [Chunk("chr1", 9000, 9500)]
        st.add(part0, np.array([0, 2120, 4240, 5040]), 0)
        st.add(part1, np.array([0, 500]), 1)
        st.add([Chunk("chr3", 0, 10)], np.array([0, 10]), None)          # a sub-batch the device could not adopt: not indexed
        seg, off = st.locate(["chr1", "chr2", "chr1", "chr1"], [1000, 150, 5100, 9000], [3120, 160, 7120, 9500])
        assert list(seg) == [0, 0, 0, 1] and list(off) == [0, 4240 + 50, 2120 + 100, 0]
        assert st.locate(["chr1"], [3000], [3200]) is None               # runs past the end of its chunk
        assert st.locate(["chr1"], [4000], [4100]) is None               # between two chunks
        assert st.locate(["chr1"], [900], [1100]) is None                # starts before the first chunk
        assert st.locate(["chr3"], [0], [5]) is None and st.locate(["chrX"], [0], [5]) is None
        assert st.locate(["chr1", "chr1"], [1000, 4000], [1010, 4010]) is None      # one uncovered region spoils the request
        assert occstore.slot_of("/x/s.occ.lower_bound.bedgraph.gz") == (1, "/x/s.occ.bedgraph.gz")
        assert occstore.slot_of("/x/s.occ.upper_bound.bedgraph.gz") == (2, "/x/s.occ.bedgraph.gz")
        assert occstore.slot_of("/x/s.occ.bedgraph.gz") == (0, "/x/s.occ.bedgraph.gz")
        occstore.register("/tmp/natac_test.occ.bedgraph.gz", st)
        assert occstore.lookup("/tmp/natac_test.occ.bedgraph.gz") is st and occstore.lookup("/tmp/other.occ.bedgraph.gz") is None
    finally:
        occstore.release("/tmp/natac_test.occ.bedgraph.gz")
    assert occstore.lookup("/tmp/natac_test.occ.bedgraph.gz") is None


def test_bench_gpus_flag_means_ranks(tmp_path):
    """bench.py --gpus N: a plain start with N > 1 spawns N ranks, a launcher's WORLD_SIZE must agree with it (exit 2 otherwise), a
    launcher without --gpus is taken at its word"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    assert bench.resolve_ranks(None, {}) == ("run", 1) and bench.resolve_ranks(1, {}) == ("run", 1)
    assert bench.resolve_ranks(8, {}) == ("spawn", 8) and bench.resolve_ranks(2, {"WORLD_SIZE": "2"}) == ("run", 2)
    assert bench.resolve_ranks(None, {"WORLD_SIZE": "4"}) == ("run", 4)
    assert bench.resolve_ranks(8, {"WORLD_SIZE": "1"})[0] == "error" and bench.resolve_ranks(1, {"WORLD_SIZE": "8"})[0] == "error"
    assert bench.resolve_ranks(0, {})[0] == "error"
    env = dict(os.environ, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 2 and "WORLD_SIZE=4" in out.stderr and not out.stdout.strip()


def test_profile_stamp_covers_the_workload(tmp_path):
    """bench.py's `traffic_source.stale` must flip when the INPUT the committed counters were collected on changes,
    not only when the kernels do: the stamp hashes nucleoatac_amd/synth.py and bench.py's workload builder too."""
    import os
    import shutil
    from nucleoatac_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = _lib.profile_sha16()
    assert base == _lib.profile_sha16() and len(base) == 16
    synth = tmp_path / "synth.py"
    shutil.copy(os.path.join(root, "nucleoatac_amd", "synth.py"), synth)
    assert _lib.profile_sha16(synth_path=str(synth)) == base                 # a copy: the same stamp
    with open(synth, "a") as f:
        f.write("\n# another fragment-size mixture\n")
    assert _lib.profile_sha16(synth_path=str(synth)) != base                 # the generator changed: stale
    bench = tmp_path / "bench.py"
    src = open(os.path.join(root, "bench.py")).read()
    assert "F = a.frags_per_chunk or 500" in src
    bench.write_text(src.replace("F = a.frags_per_chunk or 500", "F = a.frags_per_chunk or 600"))
    assert _lib.profile_sha16(bench_path=str(bench)) != base                 # the workload's constants changed: stale
    bench.write_text(src.replace("def pmc_source():", "def pmc_source():  # edited"))
    assert _lib.profile_sha16(bench_path=str(bench)) == base                 # the rest of bench.py is not the workload
    # and bench.py reports it
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bm)
    src_info = bm.pmc_source()
    assert src_info is None or src_info["current_source_sha16"] == base
