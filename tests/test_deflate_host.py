"""CPU: the host restatement of the device BGZF encoder (natac_bgzf_lines_host, csrc/natac_deflate.hpp) -- the kernels must
produce exactly its bytes (tests/test_gpu_textz.py), so here it is pinned against zlib's inflate: every member it emits must
decompress to the text, for float tracks, integer tracks, long names, single lines, members that fall back to `stored`."""
import gzip
import io
import zlib

import numpy as np

from nucleoatac_amd.pyatac.tracks import _py2_float_str as f2s
from nucleoatac_amd.writer import BGZF_EOF, bgzf_lines_host


def _track_text(kind, n, seed=0, chrom="chr12", pos=123456700):
    rng = np.random.default_rng(seed)
    if kind == "occ":
        v = np.clip(0.5 + 0.3 * np.sin(np.arange(n) / 50.0) + rng.normal(0, 0.001, n), 0, 1)
    elif kind == "norm":
        v = rng.normal(0, 0.3, n) * np.exp(rng.normal(0, 2, n))
    else:
        v = rng.poisson(0.3, n).astype(float)
    lines, a = [], 0
    while a < n:
        b = a + 1
        while b < n and v[b] == v[a]:
            b += 1
        lines.append("%s\t%d\t%d\t%s\n" % (chrom, pos + a, pos + b, f2s(float(v[a]))))
        a = b
    return "".join(lines).encode(), len(lines)


def _inflate(members):
    return gzip.GzipFile(fileobj=io.BytesIO(members + BGZF_EOF)).read()


def test_members_inflate_to_the_text_and_beat_zlib_on_float_tracks():
    for kind, better in (("occ", True), ("norm", True), ("ins", False)):
        text, nl = _track_text(kind, 40000)
        z = bgzf_lines_host(text)
        assert _inflate(z) == text, kind
        per_line, zl = len(z) / nl, len(zlib.compress(text, 4)) / nl
        print("%s: %.2f bytes per line (zlib level 4: %.2f)" % (kind, per_line, zl))
        assert per_line < zl * (1.0 if better else 1.15), (kind, per_line, zl)


def test_member_framing():
    """BGZF framing (SAM spec 4.1): 'BC' extra field, BSIZE, CRC-32 + ISIZE trailer, <= 0xff00 input bytes per member"""
    import struct
    text, _ = _track_text("occ", 6000)
    z = bgzf_lines_host(text)
    o, total, n = 0, 0, 0
    while o < len(z):
        assert z[o:o + 4] == b"\x1f\x8b\x08\x04" and z[o + 12:o + 16] == b"BC\x02\x00"
        bsize = struct.unpack_from("<H", z, o + 16)[0] + 1
        crc, isz = struct.unpack_from("<II", z, o + bsize - 8)
        raw = zlib.decompress(z[o + 18:o + bsize - 8], -15)
        assert len(raw) == isz <= 0xff00 and zlib.crc32(raw) == crc
        total += isz
        o += bsize
        n += 1
    assert o == len(z) and total == len(text) and n == (len(text) + 0xff00 - 1) // 0xff00


def test_edge_cases():
    for text in (b"chr1\t0\t1\t0.5\n",                                               # one line
                 b"c\t0\t1\t1.0\nc\t1\t2\t1.0\n" * 5000,                           # very repetitive
                 b"".join(b"scaffold_%d_random\t%d\t%d\t%g\n" % (i % 7, i * 3, i * 3 + 2, i * 0.37) for i in range(30000)),
                 b"x\t1\t2\t3.0\n" + bytes(np.random.default_rng(1).integers(33, 127, 70000).astype(np.uint8)) + b"\n"):
        assert _inflate(bgzf_lines_host(text)) == text
    # incompressible text: members fall back to `stored` blocks and still fit in 64 KiB
    rng = np.random.default_rng(2)
    noise = b"".join(bytes(rng.integers(0, 256, 50).astype(np.uint8)).replace(b"\n", b" ") + b"\n" for _ in range(4000))
    z = bgzf_lines_host(noise)
    assert _inflate(z) == noise
    assert bgzf_lines_host(b"") == b""
