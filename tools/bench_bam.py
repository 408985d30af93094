"""Throughput of the native BAM extractor (csrc/natac_bam.hpp: natac_bam_open decodes every record, keeps the forward reads of
proper pairs) on a synthetic coordinate-sorted BAM of N paired-end records with realistic record sizes (50-base reads, names of
~20 characters, one 50M cigar; BGZF members of 0xff00 bytes at zlib level 6 like samtools):  python tools/bench_bam.py 4000000"""
import os
import struct
import sys
import tempfile
import time
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def synth_bam(path, n, n_refs=4, ref_len=50_000_000, seed=0):
    rng = np.random.default_rng(seed)
    text = b"@HD\tVN:1.0\tSO:coordinate\n"
    head = [b"BAM\x01" + struct.pack("<i", len(text)) + text + struct.pack("<i", n_refs)]
    for r in range(n_refs):
        nm = ("chr%d" % (r + 1)).encode() + b"\0"
        head.append(struct.pack("<i", len(nm)) + nm + struct.pack("<i", ref_len))
    seq_len, name_len = 50, 20
    rec = np.dtype([("bs", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("lname", "u1"), ("mapq", "u1"), ("bin", "<u2"), ("ncig", "<u2"),
                    ("flag", "<u2"), ("lseq", "<i4"), ("nref", "<i4"), ("npos", "<i4"), ("tlen", "<i4"), ("name", "S%d" % name_len),
                    ("cigar", "<u4"), ("seq", "u1", (seq_len + 1) // 2), ("qual", "u1", seq_len)])
    a = np.zeros(n, dtype=rec)
    a["bs"] = rec.itemsize - 4
    a["ref"] = np.sort(rng.integers(0, n_refs, n))
    pos = rng.integers(0, ref_len - 1000, n)
    order = np.lexsort((pos, a["ref"]))
    a["pos"] = pos[order]
    a["lname"], a["mapq"], a["ncig"], a["lseq"] = name_len, 30, 1, seq_len
    fwd = rng.random(n) < 0.5
    a["flag"] = np.where(fwd, 99, 147)                    # proper pair, first forward / second reverse
    a["nref"] = a["ref"]
    tl = rng.integers(40, 600, n)
    a["tlen"] = np.where(fwd, tl, -tl)
    a["npos"] = np.maximum(a["pos"] + np.where(fwd, tl - seq_len, -(tl - seq_len)), 0)
    a["name"] = np.char.add(b"read", np.arange(n).astype("S15"))
    a["cigar"] = seq_len << 4
    a["seq"] = rng.integers(0, 256, (n, (seq_len + 1) // 2), dtype=np.uint8)
    a["qual"] = rng.integers(20, 41, (n, seq_len), dtype=np.uint8)
    data = b"".join(head) + a.tobytes()

    def member(o):
        chunk = data[o:o + 0xff00]
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        comp = co.compress(chunk) + co.flush()
        return (bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0]) + struct.pack("<H", 18 + len(comp) + 8 - 1) + comp +
                struct.pack("<II", zlib.crc32(chunk) & 0xffffffff, len(chunk)))
    with ThreadPoolExecutor(os.cpu_count() or 4) as pool, open(path, "wb") as f:
        for m in pool.map(member, range(0, len(data), 0xff00)):
            f.write(m)
        f.write(bytes([0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 0x42, 0x43, 2, 0, 0x1b, 0, 3, 0, 0, 0, 0, 0, 0, 0, 0, 0]))
    return len(data), int(fwd.sum())


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    d = tempfile.mkdtemp(prefix="natac_bam_")
    path = os.path.join(d, "synth.bam")
    try:
        raw, kept = synth_bam(path, n)
        size = os.path.getsize(path)
        from nucleoatac_amd.device import Context
        modes = [(1, False), (4, False), (0, False)] + ([(0, True), (0, True)] if Context.device_count() > 0 else [])
        for threads, device in modes:
            t0 = time.perf_counter()
            st = FragmentStore.from_bam(path, n_threads=threads, device=device)
            dt = time.perf_counter() - t0
            if device:
                threads = "device" if FragmentStore.last_bam_on_device else "device->host"
                import ctypes as C
                from nucleoatac_amd import _lib as L, get_context
                h = C.c_void_p()
                t1 = time.perf_counter()
                L.check(L.load().natac_bam_open_device(get_context()._h, path.encode(), C.byref(h), None))
                print("   natac_bam_open_device alone: %.2f s" % (time.perf_counter() - t1))
                L.load().natac_bam_close(h)
            total = sum(len(st.pos[c]) for c in st.pos) if hasattr(st, "pos") else -1
            print("threads=%s  %d records, %d kept (expected %d): %.2f s = %.1f M records/s, %.0f MB/s compressed, %.0f MB/s inflated"
                  % (threads or "auto", n, total, kept, dt, n / dt / 1e6, size / dt / 1e6, raw / dt / 1e6))
    finally:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
