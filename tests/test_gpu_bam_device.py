"""BAM -> fragment arrays on the device (csrc/natac_bam_dev.hpp: one lane inflates one BGZF member, one lane walks the records
of one member, the host confirms the chain of record starts) against the host decoder (natac_bam.hpp, itself checked against
an independent Python decoder in tests/test_bam.py): the same per-reference arrays for members of every size, windows small
enough that records and the header straddle them, every deflate block type, and the same errors for damaged files."""
import os
import struct
import zlib

import numpy as np
import pytest

from nucleoatac_amd.pyatac.fragments import FragmentStore

pytestmark = pytest.mark.gpu


def _bgzf(data, blk, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    out = bytearray()
    for o in range(0, len(data), blk):
        chunk = data[o:o + blk]
        co = zlib.compressobj(level, zlib.DEFLATED, -15, 8, strategy)
        comp = co.compress(chunk) + co.flush()
        out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", 18 + len(comp) + 8 - 1)
        out += comp + struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk))
    out += bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0])
    return bytes(out)


def _random_bam_bytes(rng, n, n_refs=5, long_header=False):
    """uncompressed BAM: records with names, cigars, sequences and aux data of every length, mapped / unmapped, every flag mix"""
    text = b"@HD\tVN:1.0\tSO:coordinate\n"
    if long_header:              # ~300 kB that do not compress
        text += b"@CO\t" + bytes(rng.integers(48, 123, 300000, dtype=np.uint8)) + b"\n"
    parts = [b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", n_refs)]
    for r in range(n_refs):
        nm = ("chr%d_%s" % (r, "y" * r)).encode() + b"\0"
        parts.append(struct.pack("<i", len(nm)) + nm + struct.pack("<i", 1 << 28))
    ref = np.sort(rng.integers(0, n_refs, n))
    ref[n - n // 50:] = -1                                   # unmapped reads at the end, like a sorted file
    pos = rng.integers(0, 1 << 27, n)
    order = np.lexsort((pos, np.where(ref < 0, n_refs, ref)))
    ref, pos = ref[order], pos[order]
    flags = rng.choice([99, 147, 83, 163, 4, 77, 141, 0, 1, 3, 1187, 2115], n)
    for i in range(n):
        ln, nc, ls, aux = int(rng.integers(2, 40)), int(rng.integers(0, 6)), int(rng.integers(0, 200)), int(rng.integers(0, 50))
        name = bytes(rng.integers(33, 127, ln - 1, dtype=np.uint8)) + b"\0"
        body = name + bytes(rng.integers(0, 256, 4 * nc + (ls + 1) // 2 + ls + aux, dtype=np.uint8))
        tlen = int(rng.integers(-700, 700))
        rec = struct.pack("<iiBBHHHiiii", int(ref[i]), int(pos[i]) if ref[i] >= 0 else -1, ln, 30, 4680, nc, int(flags[i]), ls,
                          int(ref[i]), int(pos[i]) + 40 if ref[i] >= 0 else -1, tlen) + body
        parts.append(struct.pack("<i", len(rec)) + rec)
    return b"".join(parts)


def _same(a, b):
    assert a.references == b.references and list(a.lengths) == list(b.lengths)
    for c in a.references:
        assert np.array_equal(a.pos[c], b.pos[c]) and np.array_equal(a.tlen[c], b.tlen[c]), c


@pytest.mark.parametrize("blk,level,window", [(300, 6, 0), (3000, 1, 0), (65280, 6, 0), (65280, 9, 40000), (5000, 0, 0), (777, 6, 2500),
                                              (65536, 6, 0), (1200, 6, 70000)])
def test_device_decoder_equals_host_decoder(tmp_path, monkeypatch, blk, level, window):
    rng = np.random.default_rng(blk + level)
    raw = _random_bam_bytes(rng, 6000, long_header=(blk == 1200))
    path = str(tmp_path / "a.bam")
    strategy = zlib.Z_FIXED if blk == 777 else zlib.Z_DEFAULT_STRATEGY
    open(path, "wb").write(_bgzf(raw, blk, level, strategy))
    if window:
        monkeypatch.setenv("NATAC_BAM_DEV_WINDOW", str(window))
    host = FragmentStore.from_bam(path, device=False)
    dev = FragmentStore.from_bam(path, device=True)
    if blk == 1200:
        # a header that does not fit the (70,000 + 65,536)-byte window: the device path hands the file to the host decoder
        assert FragmentStore.last_bam_on_device is False
    else:
        assert FragmentStore.last_bam_on_device is True
    _same(host, dev)
    assert sum(len(host.pos[c]) for c in host.references) > 1000


def test_device_decoder_large_file_and_damage(tmp_path):
    """400,000 records through 64-KiB members (several thousand lanes), then the same file truncated / garbled: the device path
    reports what the host decoder reports"""
    from helpers import write_bam
    rng = np.random.default_rng(5)
    n = 400000
    ref = np.sort(rng.integers(0, 3, n))
    pos = rng.integers(0, 5_000_000, n)
    order = np.lexsort((pos, ref))
    flag = rng.choice([99, 147, 83, 163], n)
    tl = rng.integers(30, 900, n) * np.where(flag & 0x10, -1, 1)
    path = str(tmp_path / "big.bam")
    write_bam(path, [("chrI", 6_000_000), ("chrII", 6_000_000), ("chrIII", 6_000_000)],
              zip(ref[order].tolist(), pos[order].tolist(), flag.tolist(), tl.tolist()), blk=65280)
    host = FragmentStore.from_bam(path, device=False)
    dev = FragmentStore.from_bam(path, device=True)
    assert FragmentStore.last_bam_on_device is True
    _same(host, dev)
    assert sum(len(dev.pos[c]) for c in dev.references) == int(((flag & 2) > 0).sum() - ((flag & 0x12) == 0x12).sum())
    b = open(path, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(b[:len(b) // 2])
    with pytest.raises(Exception, match="truncated|trailing"):
        FragmentStore.from_bam(cut, device=True)
    g = bytearray(b)
    g[len(g) // 3] ^= 0x55
    g[len(g) // 3 + 1] ^= 0xaa
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(g))
    for device in (False, True):       # a damaged deflate stream either fails to inflate or fails its member's CRC-32
        with pytest.raises(Exception, match="inflate|BGZF|truncated"):
            FragmentStore.from_bam(bad, device=device)


def test_payload_damage_that_still_inflates_fails_the_crc(tmp_path):
    """members written with stored deflate blocks: a flipped bit in the data bytes inflates without complaint to the right length --
    only the member's CRC-32 (which htslib verifies behind every read of the reference, pyatac/fragments.pyx:21) can tell; both
    decoders name the member's file offset"""
    rng = np.random.default_rng(17)
    raw = _random_bam_bytes(rng, 3000)
    blk = 5000
    good = _bgzf(raw, blk, 0)                                   # level 0: one stored block per member
    path = str(tmp_path / "stored.bam")
    open(path, "wb").write(good)
    _same(FragmentStore.from_bam(path, device=False), FragmentStore.from_bam(path, device=True))
    # member k starts at k * (18 + 5 + blk + 8); its data bytes follow the 5-byte stored-block header
    per = 18 + 5 + blk + 8
    for k, byte in ((0, 40), (3, 4999), (len(raw) // blk - 1, 123)):
        g = bytearray(good)
        g[k * per + 18 + 5 + byte] ^= 0x04
        bad = str(tmp_path / ("flip%d.bam" % k))
        open(bad, "wb").write(bytes(g))
        for device in (False, True):
            with pytest.raises(Exception, match=r"CRC-32 mismatch in the BGZF member at file offset %d " % (k * per)):
                FragmentStore.from_bam(bad, device=device)
    # a damaged CRC field itself
    g = bytearray(good)
    g[2 * per + 18 + 5 + blk] ^= 0x80
    bad = str(tmp_path / "crcfield.bam")
    open(bad, "wb").write(bytes(g))
    for device in (False, True):
        with pytest.raises(Exception, match=r"CRC-32 mismatch in the BGZF member at file offset %d " % (2 * per)):
            FragmentStore.from_bam(bad, device=device)
