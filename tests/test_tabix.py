"""natac_bgzip_file / natac_tabix_index (csrc/natac_tabix.hpp) + the TabixFile reader: region queries through the index
must return exactly what a linear scan of the text returns (pysam.tabix_index / TabixFile.fetch of the reference)."""
import gzip
import os
import struct

import numpy as np
import pytest

from nucleoatac_amd import _lib as L
from nucleoatac_amd.tabix import TabixFile, reg2bins
from nucleoatac_amd.writer import bgzip_file, tabix_index, write_bedgraph


def _brute(lines, chrom, start, end):
    out = []
    for ln in lines:
        f = ln.split("\t")
        if f[0] == chrom and int(f[1]) < end and max(int(f[2]), int(f[1]) + 1) > start:
            out.append(ln)
    return out


def _random_bed(rng, n_per_chrom, chroms, span):
    lines = []
    for c in chroms:
        pos = np.sort(rng.integers(0, span, n_per_chrom))
        for p in pos:
            lines.append("%s\t%d\t%d\t%.3f" % (c, p, p + int(rng.integers(1, 400)), rng.random()))
    return lines


def test_reg2bins_contains_reg2bin():
    rng = np.random.default_rng(0)
    for _ in range(2000):
        b = int(rng.integers(0, 1 << 29) - 1000)
        b = max(b, 0)
        e = b + int(rng.integers(1, 1 << int(rng.integers(1, 27))))
        e = min(e, 1 << 29)
        if e <= b:
            continue
        # bin of the record itself (same formula as csrc reg2bin)
        ee = e - 1
        for shift, off in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
            if b >> shift == ee >> shift:
                rb = off + (b >> shift)
                break
        else:
            rb = 0
        assert rb in reg2bins(b, e)


def test_bgzip_and_index_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    lines = ["# a comment line"] + _random_bed(rng, 30000, ["chrI", "chrII", "chrX_random"], 3_000_000)
    src = tmp_path / "peaks.bed"
    src.write_text("\n".join(lines) + "\n")
    gz = bgzip_file(str(src), level=4)
    assert gz.endswith(".bed.gz") and not os.path.exists(str(src))
    with gzip.open(gz, "rt") as fh:               # plain gzip readers see the same text
        assert fh.read().split("\n")[:-1] == lines
    with open(gz, "rb") as fh:
        assert fh.read()[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 66, 67, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    n = tabix_index(gz)
    assert n == len(lines) - 1
    with gzip.open(gz + ".tbi", "rb") as fh:
        raw = fh.read()
    assert raw[:4] == b"TBI\x01"
    n_ref, fmt, cs, cb, ce, meta, skip, l_nm = struct.unpack_from("<8i", raw, 4)
    assert (n_ref, fmt, cs, cb, ce, meta, skip) == (3, 0x10000, 1, 2, 3, ord("#"), 0)
    tb = TabixFile(gz)
    assert tb.contigs == ["chrI", "chrII", "chrX_random"]
    assert [int(s[1][0]) for s in tb.stats] == [30000, 30000, 30000]
    body = lines[1:]
    for _ in range(300):
        c = tb.contigs[int(rng.integers(0, 3))]
        s = int(rng.integers(0, 3_000_000))
        e = s + int(rng.integers(1, 1 << int(rng.integers(1, 21))))
        assert list(tb.fetch(c, s, e)) == _brute(body, c, s, e), (c, s, e)
    assert list(tb.fetch("chrI", 0, 1 << 29)) == [ln for ln in body if ln.startswith("chrI\t")]
    assert list(tb.fetch("nope", 0, 100)) == []
    tb.close()


def test_index_of_native_bedgraph_writer(tmp_path):
    """the multi-threaded track writer's BGZF output is indexable, and queries agree with the values written"""
    rng = np.random.default_rng(2)
    nc = 40
    lens = rng.integers(500, 30000, nc)
    chroms = ["chr1"] * 25 + ["chr2"] * 15
    starts = np.concatenate([np.cumsum(np.r_[100, lens[:24] + 50]), np.cumsum(np.r_[5000, lens[25:39] + 70])])
    off = np.r_[0, np.cumsum(lens)]
    vals = np.round(rng.random(int(off[-1])) * 4) / 4.0          # runs of equal values -> run-length lines
    path = str(tmp_path / "t.bedgraph.gz")
    write_bedgraph(path, chroms, starts, off, vals, compress=4)
    n = tabix_index(path)
    with gzip.open(path, "rt") as fh:
        body = fh.read().split("\n")[:-1]
    assert n == len(body)
    tb = TabixFile(path)
    for _ in range(200):
        i = int(rng.integers(0, nc))
        s = int(starts[i] + rng.integers(0, lens[i]))
        e = s + int(rng.integers(1, 3000))
        got = list(tb.fetch(chroms[i], s, e))
        assert got == _brute(body, chroms[i], s, e)
        # and the lines reproduce the written values on the queried bases of this chunk
        for ln in got:
            c, b0, e0, v = ln.split("\t")
            for x in range(max(int(b0), s), min(int(e0), e, int(starts[i] + lens[i]))):
                if starts[i] <= x:
                    assert float(v) == vals[off[i] + x - starts[i]]
    tb.close()


def test_index_rejects_unsorted_and_plain_gzip(tmp_path):
    src = tmp_path / "u.bed"
    src.write_text("chr1\t500\t600\nchr1\t100\t200\n")
    gz = bgzip_file(str(src))
    with pytest.raises(L.NatacError):
        tabix_index(gz)
    plain = str(tmp_path / "p.bed.gz")
    with gzip.open(plain, "wt") as fh:
        fh.write("chr1\t1\t2\n")
    with pytest.raises(L.NatacError):
        tabix_index(plain)
    split = tmp_path / "s.bed"
    split.write_text("chr1\t1\t2\nchr2\t1\t2\nchr1\t5\t6\n")
    with pytest.raises(L.NatacError):
        tabix_index(bgzip_file(str(split)))


def test_read_track_through_index_equals_linear_scan(tmp_path):
    """Track.read_track gives the same values through the tabix index (bulk parse) as through a plain linear scan"""
    import shutil
    from nucleoatac_amd.pyatac.tracks import Track
    rng = np.random.default_rng(4)
    lens = rng.integers(800, 6000, 12)
    chroms = ["chr1"] * 7 + ["chr2"] * 5
    starts = np.concatenate([np.cumsum(np.r_[300, lens[:6] + 200]), np.cumsum(np.r_[900, lens[7:11] + 90])])
    off = np.r_[0, np.cumsum(lens)]
    vals = rng.random(int(off[-1]))
    vals[rng.integers(0, len(vals), 500)] = np.nan                       # NaN runs are not written
    vals[1000:1400] = 0.25                                               # a multi-base run
    path = str(tmp_path / "t.bedgraph.gz")
    write_bedgraph(path, chroms, starts, off, vals, compress=4)
    noidx = str(tmp_path / "n.bedgraph.gz")
    shutil.copy(path, noidx)
    tabix_index(path)
    for _ in range(40):
        i = int(rng.integers(0, 12))
        s = int(starts[i] + rng.integers(-50, lens[i]))
        e = s + int(rng.integers(1, 2500))
        a, b = Track(chroms[i], s, e), Track(chroms[i], s, e)
        a.read_track(path)                                             # native reader (natac_tbx_read_values)
        b.read_track(noidx)
        assert np.array_equal(a.vals, b.vals, equal_nan=True)
        from nucleoatac_amd.tabix import TabixFile
        tb = TabixFile(path)                                           # pure-Python reader, bulk parse
        b0, e0, v0 = tb.fetch_values(chroms[i], max(0, s), e)
        c = np.full(e - s, np.nan)
        for x0, x1, v in zip(b0, e0, v0):
            c[max(x0 - s, 0):min(x1 - s, e - s)] = v
        tb.close()
        assert np.array_equal(a.vals, c, equal_nan=True)
    t = Track("chrNone", 5, 50)
    t.read_track(path, empty=-1.0)
    assert (t.vals == -1.0).all()
    t = Track("chr1", -40, 30)                                           # negative start: bases before 0 stay empty
    t.read_track(path)
    assert np.isnan(t.vals).all()


@pytest.mark.parametrize("blk", [37, 700, 5000, 65280])
def test_region_reads_with_cached_members_equal_a_linear_scan(tmp_path, blk):
    """The native reader keeps the last inflated members with their lines split (natac_tabix.hpp: Reader::Block).  Members
    that end anywhere -- in the middle of a line, of a number, with no newline at all (37-byte members) -- and reads in
    ascending, repeated and random order must give what a scan of the text gives; also the record-start column of a BED file
    (value_col 2, the dyads of `nfr`), comment lines and overlapping records (later records overwrite earlier ones)."""
    from helpers import bgzf_bytes
    from nucleoatac_amd.tabix import NativeTabix
    rng = np.random.default_rng(blk)
    recs = []
    for c, n in (("chrA", 3000), ("chrB", 40), ("chrC_long_name", 1500)):
        pos = np.sort(rng.integers(0, 60000, n))
        for p in pos:
            recs.append((c, int(p), int(p) + int(rng.choice([1, 1, 1, 7, 300, 20000], p=[.5, .2, .1, .1, .09, .01])), float(rng.normal())))
    text = "#comment line\n" + "".join("%s\t%d\t%d\t%r\n" % r for r in recs)
    path = str(tmp_path / "r.bed.gz")
    open(path, "wb").write(bgzf_bytes(text.encode(), blk))
    tabix_index(path)

    def brute(chrom, s, e, col):
        out = np.full(e - s, np.nan)
        for c, b0, e0, v in recs:
            if c == chrom and b0 < e and e0 > max(s, 0):
                out[max(b0, s) - s:min(e0, e) - s] = v if col == 4 else b0
        return out

    rd = NativeTabix(path)
    regions = [(c, s, s + w) for c in ("chrA", "chrC_long_name", "chrB") for s, w in zip(range(0, 60000, 4100), [2120] * 15)]
    regions += [("chrA", int(s), int(s) + int(w)) for s, w in zip(rng.integers(-50, 61000, 40), rng.integers(1, 9000, 40))]
    regions += regions[:10]                                             # again, now from the cache
    for chrom, s, e in regions:
        for col in (4, 2):
            assert np.array_equal(rd.read_values(chrom, s, e, value_col=col), brute(chrom, s, e, col), equal_nan=True), (chrom, s, e, col)
    assert np.isnan(rd.read_values("chrNone", 0, 100)).all()
    rd.close()


def test_read_regions_equals_single_reads_and_parses_numbers_like_float(tmp_path):
    """natac_tbx_read_regions (many regions per call, contiguous runs per cursor) against read_values per region; the value
    parser (one exact multiplication / division for <= 15 digits and |exponent| <= 22, strtod otherwise) against python's float()
    on 12-digit track values, reprs, exponent forms, integers, nan, inf, denormals"""
    from helpers import bgzf_bytes
    from nucleoatac_amd.tabix import NativeTabix
    rng = np.random.default_rng(11)
    n = 120000
    x = rng.normal(size=n) * 10.0 ** rng.integers(-30, 30, n)
    forms = ["%.12g", "%r", "%e", "%.15g", "%.16g", "%d", "%.3f", "%.12g"]
    texts = []
    for i, v in enumerate(x):
        f = forms[i % len(forms)]
        texts.append(f % (int(v % 1e15) if f == "%d" else float(v)))
    special = ["nan", "inf", "-inf", "-0.0", "0", "1e22", "1e23", "1e-22", "1e-23", "123456789012345", "1234567890123456", "4.9e-324",
               "2.2250738585072014e-308", "1.7976931348623157e308", "+5.5", "0.000000000000000000001", "9007199254740993", ".5", "5.",
               "1E5", "12345678901234567890123", "0.1e-21", "1e+5"]
    texts[:len(special)] = special
    text = "".join("c1\t%d\t%d\t%s\n" % (i, i + 1, t) for i, t in enumerate(texts))
    path = str(tmp_path / "v.bed.gz")
    open(path, "wb").write(bgzf_bytes(text.encode(), 60000))
    tabix_index(path)
    rd = NativeTabix(path)
    want = np.array([float(t) for t in texts])
    starts = np.arange(0, n, 997)
    ends = np.minimum(starts + 997, n)
    flat, off = rd.read_regions(["c1"] * len(starts), starts, ends, n_threads=5)
    assert np.array_equal(off, np.r_[0, np.cumsum(ends - starts)])
    same = (flat == want) | (np.isnan(flat) & np.isnan(want))
    assert same.all(), [texts[i] for i in np.flatnonzero(~same)[:10]]
    assert np.array_equal(np.signbit(flat), np.signbit(want))
    # arbitrary (overlapping, unsorted, empty, other-chromosome) region lists: the same values as one read per region
    chroms = ["c1"] * 60 + ["cX"] * 3
    s2 = np.r_[rng.integers(-100, n, 60), [5, 10, 15]]
    e2 = s2 + np.r_[rng.integers(0, 5000, 60), [100, 0, 7]]
    for threads in (1, 4):
        flat, off = rd.read_regions(chroms, s2, e2, n_threads=threads, empty=-7.0)
        for i in range(len(chroms)):
            assert np.array_equal(flat[off[i]:off[i + 1]], rd.read_values(chroms[i], int(s2[i]), int(e2[i]), empty=-7.0), equal_nan=True), i
    flat, off = rd.read_regions([], [], [])
    assert len(flat) == 0 and list(off) == [0]
    rd.close()
