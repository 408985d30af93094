"""Randomised sweep of the device BAM decoder (natac_bam_open_device) against the host decoder: random record mixes (names, cigars,
sequences, aux data of every length, every flag combination, unmapped reads), member sizes from 64 bytes to 64 KiB, deflate levels and
strategies (stored / fixed / dynamic blocks), device windows from a few kB up.   usage: python tests/fuzz/fuzz_bam.py [rounds] [seed]"""
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from nucleoatac_amd.pyatac.fragments import FragmentStore     # noqa: E402
import test_gpu_bam_device as T                                # noqa: E402


def run(rounds, seed):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    d = tempfile.mkdtemp(prefix="natac_fuzz_bam_")
    path = os.path.join(d, "f.bam")
    records = on_device = done = 0
    for r in range(rounds):
        n = int(rng.integers(50, 4000))
        raw = T._random_bam_bytes(rng, n, n_refs=int(rng.integers(1, 9)))
        blk = int(rng.choice([64, 200, 1000, 4096, 20000, 65280, 65536]))
        level = int(rng.choice([0, 1, 6, 9]))
        strategy = rng.choice([zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE])
        if blk > 50000 and (level == 0 or strategy != zlib.Z_DEFAULT_STRATEGY):
            blk = 50000                      # random bytes do not shrink that way: the member must stay below 64 KiB
        open(path, "wb").write(T._bgzf(raw, blk, level, int(strategy)))
        window = int(rng.choice([0, 3000, 70000, 1 << 20]))
        if window:
            os.environ["NATAC_BAM_DEV_WINDOW"] = str(window)
        else:
            os.environ.pop("NATAC_BAM_DEV_WINDOW", None)
        host = FragmentStore.from_bam(path, device=False)
        dev = FragmentStore.from_bam(path, device=True)
        on_device += bool(FragmentStore.last_bam_on_device)
        T._same(host, dev)
        records += n
        done = r + 1
        if time.time() - t0 > float(os.environ.get("FUZZ_SECONDS", "1e9")):
            break
    os.environ.pop("NATAC_BAM_DEV_WINDOW", None)
    import shutil
    shutil.rmtree(d, ignore_errors=True)
    print("bam fuzz ok: %d rounds (%d answered by the device), %d records, %.0f s" % (done, on_device, records, time.time() - t0))
    return done, on_device


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
