"""CPU: native BAM extractor (natac_bam_*, host code in libnatac_hip.so) vs the pure-Python decoder vs ground truth,
on a synthetic multi-block BAM written here and on the reference's own single_read.bam fixture."""
import os
import struct
import zlib

import numpy as np
import pytest

from helpers import GOLDEN, golden
from nucleoatac_amd.pyatac.fragments import FragmentStore


from helpers import bgzf_bytes as _bgzf, write_bam as _write_bam  # noqa: E402


def test_native_decoder_matches_truth_and_python(tmp_path):
    rng = np.random.default_rng(0)
    refs = [("chrA", 50000), ("chrB", 80000), ("chrEmpty", 1000)]
    recs = []
    for ref_id, ln in ((0, 50000), (1, 80000)):
        pos = np.sort(rng.integers(0, ln - 1000, size=4000))
        for p in pos:
            flag = int(rng.choice([0x63, 0x53, 0x93, 0xa3, 0x41, 0x4, 0x163]))   # fwd proper, rev proper, improper, unmapped, ...
            tl = int(rng.integers(30, 700)) * (1 if not (flag & 0x10) else -1)
            recs.append((ref_id, int(p), flag, tl))
    recs.append((-1, -1, 0x4, 0))                                             # unplaced read
    path = str(tmp_path / "synth.bam")
    _write_bam(path, refs, recs)
    nat = FragmentStore.from_bam(path)
    py = FragmentStore.from_bam_python(path)
    assert nat.references == ["chrA", "chrB", "chrEmpty"] and nat.lengths == [50000, 80000, 1000]
    for i, c in enumerate(["chrA", "chrB"]):
        keep = [(p, abs(t)) for (r, p, f, t) in recs if r == i and (f & 0x2) and not (f & 0x10)]
        assert len(keep) > 500
        assert np.array_equal(nat.pos[c], [k[0] for k in keep]) and np.array_equal(nat.tlen[c], [k[1] for k in keep])
        assert np.array_equal(nat.pos[c], py.pos[c]) and np.array_equal(nat.tlen[c], py.tlen[c])
    assert len(nat.pos["chrEmpty"]) == 0
    l, n = nat.fetch("chrA", 1000, 3000)
    assert np.all(np.diff(l) >= 0) and len(l) == len(n)


def test_reference_fixture_single_read():
    st = FragmentStore.from_bam(os.path.join(GOLDEN, "ref_single_read.bam"))
    sr = golden("single_read")
    l, n = st.fetch("chrII", int(sr["start"]), int(sr["end"]))
    assert np.array_equal(l, sr["l"]) and np.array_equal(n, sr["n"])
    assert len(st.references) == 17 and st.chrom_sizes()["chrII"] == 813184


def test_corrupt_inputs_fail_loudly(tmp_path):
    from nucleoatac_amd._lib import NatacError
    p = str(tmp_path / "x.bam")
    open(p, "wb").write(b"not a bam")
    with pytest.raises(NatacError):
        FragmentStore.from_bam(p)
    _write_bam(p, [("c", 100)], [(0, 5, 0x63, 100)])
    raw = open(p, "rb").read()
    open(p, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(NatacError):
        FragmentStore.from_bam(p)
    with pytest.raises(NatacError):
        FragmentStore.from_bam(str(tmp_path / "missing.bam"))


@pytest.mark.parametrize("window", ["4096", "20000", "1000000"])
def test_streaming_windows_give_identical_arrays(tmp_path, monkeypatch, window):
    """the streaming decoder with tiny compressed windows (records, the header and BGZF blocks straddle window borders)
    equals the pure-Python whole-file decoder"""
    rng = np.random.default_rng(4)
    refs = [("chr%d" % i, 100000 + i) for i in range(40)]                      # a header longer than one 3000-byte block
    recs = []
    for ref_id in (0, 7, 39):
        for p in np.sort(rng.integers(0, 90000, size=3000)):
            flag = int(rng.choice([0x63, 0x53, 0x93, 0xa3]))
            recs.append((ref_id, int(p), flag, int(rng.integers(30, 700)) * (1 if not (flag & 0x10) else -1)))
    path = str(tmp_path / "w.bam")
    _write_bam(path, refs, recs)
    monkeypatch.setenv("NATAC_BAM_WINDOW", window)
    nat = FragmentStore.from_bam(path)
    py = FragmentStore.from_bam_python(path)
    assert nat.references == py.references and nat.lengths == py.lengths
    for c in nat.references:
        assert np.array_equal(nat.pos[c], py.pos[c]) and np.array_equal(nat.tlen[c], py.tlen[c]), c
    assert sum(len(nat.pos[c]) for c in nat.references) > 3000


def test_streaming_rejects_truncated_files(tmp_path, monkeypatch):
    from nucleoatac_amd import _lib as L
    refs = [("chrA", 5000)]
    path = str(tmp_path / "t.bam")
    _write_bam(path, refs, [(0, 10 * i, 0x63, 200) for i in range(2000)])
    raw = open(path, "rb").read()
    monkeypatch.setenv("NATAC_BAM_WINDOW", "4096")
    for cut in (len(raw) - 40, len(raw) // 2, 30):
        bad = str(tmp_path / ("cut%d.bam" % cut))
        open(bad, "wb").write(raw[:cut])
        with pytest.raises(L.NatacError):
            FragmentStore.from_bam(bad)


def test_corrupt_isize_is_rejected_before_allocation(tmp_path):
    """a BGZF member whose ISIZE field claims gigabytes must be rejected ("corrupt BGZF block"), not used to size the buffer"""
    import struct
    refs = [("chrA", 50000)]
    path = str(tmp_path / "ok.bam")
    _write_bam(path, refs, [(0, 100 + i, 0x63, 200) for i in range(300)])
    raw = bytearray(open(path, "rb").read())
    bsize = struct.unpack_from("<H", raw, 16)[0] + 1                 # first member: overwrite its ISIZE with 3 GB
    struct.pack_into("<I", raw, bsize - 4, 3000000000)
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(Exception) as e:
        FragmentStore.from_bam(bad)
    assert "corrupt BGZF block" in str(e.value)


def test_device_inflate_algorithm_on_the_host():
    """natac_bam_dev.hpp's raw-deflate decoder is __host__ __device__: here it runs on the CPU against zlib on stored, fixed and
    dynamic blocks, several blocks per member, empty and maximal members; damaged input gives an error code, never a crash"""
    import ctypes as C
    import zlib
    from nucleoatac_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(0)

    def inflate(comp, n):
        out = C.create_string_buffer(max(1, n))
        return lib.natac_inflate_raw_host(comp, len(comp), out, n), out.raw[:n]

    cases = []
    for n in (0, 1, 2, 100, 3000, 65280, 65536):
        cases += [bytes(rng.integers(0, 256, n, dtype=np.uint8)), bytes(rng.integers(0, 4, n, dtype=np.uint8)),
                  (b"chr1\t12345\t12346\t0.123456789012\n" * (n // 30 + 1))[:n], bytes(n)]
    for d in cases:
        for lvl in (0, 1, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE):
                co = zlib.compressobj(lvl, zlib.DEFLATED, -15, 8, strat)
                comp = co.compress(d) + co.flush()
                rc, got = inflate(comp, len(d))
                assert rc == 0 and got == d, (len(d), lvl, strat, rc)
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    parts, data = [], b""
    for i in range(20):
        d = bytes(rng.integers(0, 1 + i * 12, 3000, dtype=np.uint8))
        data += d
        parts += [co.compress(d), co.flush(zlib.Z_FULL_FLUSH if i % 2 else zlib.Z_SYNC_FLUSH)]
    comp = b"".join(parts) + co.flush()
    assert inflate(comp, len(data)) == (0, data)
    assert inflate(comp[:len(comp) // 2], len(data))[0] != 0          # truncated input
    assert inflate(comp, len(data) - 5)[0] != 0                        # ISIZE too small
    assert inflate(comp, len(data) + 5)[0] != 0                        # ISIZE too large
    for k in range(0, len(comp), 97):                                  # flipped bytes: an error or other bytes, no crash
        g = bytearray(comp)
        g[k] ^= 0xff
        inflate(bytes(g), len(data))


def test_prefetch_hands_the_store_or_the_error_to_open(tmp_path):
    """FragmentStore.prefetch decodes on a thread; FragmentStore.open waits for it and raises what the decode raised"""
    from helpers import write_bam
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    good = str(tmp_path / "g.bam")
    write_bam(good, [("chrI", 10000)], [(0, 100, 99, 150), (0, 220, 147, -150), (0, 300, 163, 90)])
    FragmentStore.prefetch(good)
    st = FragmentStore.open(good)
    assert st.references == ["chrI"] and list(st.pos["chrI"]) == [100, 300]
    assert FragmentStore.open(good) is st                    # cached
    bad = str(tmp_path / "missing.bam")
    FragmentStore.prefetch(bad)
    with pytest.raises(Exception, match="cannot open"):
        FragmentStore.open(bad)


def test_crc_of_every_member_is_verified(tmp_path):
    """a flipped bit inside a STORED deflate block inflates to the right length: only the member's CRC-32 tells (RFC 1952; htslib
    checks it behind pyatac/fragments.pyx:21).  The host BAM decoder names the member's file offset; the tabix reader refuses the
    damaged member of a track file."""
    import struct
    import zlib
    from nucleoatac_amd._lib import NatacError

    def bgzf0(data, blk):
        out = bytearray()
        for o in range(0, len(data), blk):
            chunk = data[o:o + blk]
            co = zlib.compressobj(0, zlib.DEFLATED, -15)
            comp = co.compress(chunk) + co.flush()
            out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", 18 + len(comp) + 8 - 1)
            out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
        return bytes(out)

    p = str(tmp_path / "x.bam")
    _write_bam(p, [("c", 100000)], [(0, 5 + 7 * i, 0x63, 100 + i % 50) for i in range(2000)])
    import gzip
    raw = gzip.open(p, "rb").read()
    blk = 4000
    good = bgzf0(raw, blk)
    open(p, "wb").write(good)
    ok = FragmentStore.from_bam(p, device=False)
    assert len(ok.pos["c"]) == 2000
    per = 18 + 5 + blk + 8
    for k in (0, 2, len(raw) // blk - 1):
        g = bytearray(good)
        g[k * per + 18 + 5 + 77] ^= 0x01
        bad = str(tmp_path / ("bad%d.bam" % k))
        open(bad, "wb").write(bytes(g))
        with pytest.raises(NatacError, match=r"CRC-32 mismatch in the BGZF member at file offset %d " % (k * per)):
            FragmentStore.from_bam(bad, device=False)
    # a track file: the tabix reader must not hand out values of a member that fails its CRC
    from nucleoatac_amd.pyatac.tracks import Track
    from nucleoatac_amd.writer import tabix_index
    text = "".join("chr1\t%d\t%d\t%s\n" % (i, i + 1, repr(0.25 + i * 1e-3)) for i in range(3000)).encode()
    t = str(tmp_path / "t.bedgraph.gz")
    z = bgzf0(text, 3000)
    open(t, "wb").write(z)
    tabix_index(t)
    tr = Track("chr1", 100, 200)
    tr.read_track(t)
    assert abs(tr.vals[0] - 0.35) < 1e-12
    g = bytearray(z)
    g[18 + 5 + 1500] ^= 0x01                     # inside a digit of the first member: still text, still inflates
    t2 = str(tmp_path / "t2.bedgraph.gz")         # a new path (readers cache inflated members per file): damaged data, intact index
    open(t2, "wb").write(bytes(g))
    open(t2 + ".tbi", "wb").write(open(t + ".tbi", "rb").read())
    with pytest.raises(Exception):
        tr2 = Track("chr1", 100, 200)
        tr2.read_track(t2)
    with pytest.raises(Exception):               # and the indexer, which inflates every member, refuses the file too
        tabix_index(t2)
