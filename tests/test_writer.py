"""CPU: the native bedGraph / BGZF writer (natac_write_bedgraph, host code in libnatac_hip.so) against the Python
Track.write_track mirror of the reference's text format (pyatac/tracks.py:37-74)."""
import gzip
import io

import numpy as np
import pytest

from nucleoatac_amd.pyatac.tracks import Track
from nucleoatac_amd.writer import write_bedgraph


def _case(seed=0, nc=37, L=611):
    rng = np.random.default_rng(seed)
    vals = rng.normal(size=nc * L)
    vals[rng.random(nc * L) < 0.15] = np.nan
    vals[100:400] = 0.0
    vals[700:710] = 2.0
    vals[900:905] = [1e-7 / 3, 123456789012345.0, -0.5, np.inf, 1e22]
    vals[L - 3:L + 3] = 7.25                      # equal values across a chunk border must NOT merge
    off = np.arange(nc + 1) * L
    starts = np.arange(nc) * 5000 + 100
    chroms = ["chr%d" % (i % 3) for i in range(nc)]
    return chroms, starts, off, vals, L


def _python_text(chroms, starts, off, vals, write_zero=True):
    h = io.StringIO()
    for i in range(len(chroms)):
        n = int(off[i + 1] - off[i])
        Track(chroms[i], int(starts[i]), int(starts[i]) + n, vals=vals[off[i]:off[i + 1]]).write_track(h, write_zero=write_zero)
    return h.getvalue()


@pytest.mark.parametrize("threads", [1, 3, 0])
def test_text_identical_to_write_track(tmp_path, threads):
    chroms, starts, off, vals, L = _case()
    p = str(tmp_path / "a.bedgraph")
    n = write_bedgraph(p, chroms, starts, off, vals, n_threads=threads)
    txt = open(p).read()
    assert n == len(txt.encode()) and txt == _python_text(chroms, starts, off, vals)
    assert "chr0\t708\t711\t7.25\nchr1\t5100\t5103\t7.25\n" in txt
    write_bedgraph(p, chroms, starts, off, vals, write_zero=False)
    assert open(p).read() == _python_text(chroms, starts, off, vals, write_zero=False)


def test_bgzf_roundtrip_and_append(tmp_path):
    chroms, starts, off, vals, L = _case(seed=3, nc=300, L=400)     # > 64 KiB of text per thread: several BGZF blocks
    ref = _python_text(chroms, starts, off, vals)
    p = str(tmp_path / "a.bedgraph.gz")
    write_bedgraph(p, chroms[:120], starts[:120], off[:121], vals, compress=4, finish=False)
    write_bedgraph(p, chroms[120:], starts[120:], off[120:], vals, compress=4, append=True, finish=True)
    raw = open(p, "rb").read()
    assert raw[:4] == b"\x1f\x8b\x08\x04" and raw[12:14] == b"BC"
    assert raw[-28:] == bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    assert gzip.open(p, "rt").read() == ref
    # every member is a self-contained block of <= 64 KiB (what tabix indexes)
    o, nblk = 0, 0
    while o < len(raw):
        bsize = int.from_bytes(raw[o + 16:o + 18], "little") + 1
        assert raw[o:o + 4] == b"\x1f\x8b\x08\x04" and bsize <= 65536
        o += bsize
        nblk += 1
    assert o == len(raw) and nblk > 3


def test_empty_and_errors(tmp_path):
    p = str(tmp_path / "e.bedgraph")
    assert write_bedgraph(p, [], [], [0], np.zeros(0)) == 0 and open(p).read() == ""
    allnan = np.full(50, np.nan)
    assert write_bedgraph(p, ["c"], [0], [0, 50], allnan) == 0
    from nucleoatac_amd._lib import NatacError
    with pytest.raises(NatacError):
        write_bedgraph(str(tmp_path / "no_such_dir" / "x"), ["c"], [0], [0, 50], allnan)
    with pytest.raises(NatacError):
        write_bedgraph(p, ["c"], [0], [0, 50], allnan, compress=11)
    with pytest.raises(ValueError):
        write_bedgraph(p, ["c"], [0, 1], [0, 50], allnan)


@pytest.mark.parametrize("name", ["nucleoatac_signal", "occ", "ins"])
def test_text_matches_reference_output_files(tmp_path, name):
    """format pin against the REFERENCE's own output (example/example_results/*.bedgraph.gz, first 3000 lines):
    parse -> per-base values -> native writer reproduces the file text byte for byte (12 significant digits, run lengths)"""
    import os
    from helpers import GOLDEN
    src = os.path.join(GOLDEN, "ref_results_%s.head3000.bedgraph.gz" % name)
    ref = gzip.open(src, "rt").read()
    chunks = []          # contiguous runs of lines form one chunk each
    for line in ref.strip().split("\n"):
        c, s, e, v = line.split("\t")
        s, e, v = int(s), int(e), float(v)
        if chunks and chunks[-1][0] == c and chunks[-1][2] == s:
            chunks[-1][2] = e
            chunks[-1][3].append(np.full(e - s, v))
        else:
            chunks.append([c, s, e, [np.full(e - s, v)]])
    chroms = [c[0] for c in chunks]
    starts = [c[1] for c in chunks]
    vals = np.concatenate([np.concatenate(c[3]) for c in chunks])
    off = np.concatenate(([0], np.cumsum([c[2] - c[1] for c in chunks])))
    p = str(tmp_path / "o.bedgraph")
    write_bedgraph(p, chroms, starts, off, vals)
    assert open(p).read() == ref
