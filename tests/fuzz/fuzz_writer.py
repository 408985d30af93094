"""Randomised sweep of the device track writer (natac_batch_format_track) against the native host writer and the host restatement of
the encoder: random chunk geometries and chromosome names, tracks built from values of every scale with runs of equal values, zero
runs and NaN stretches of every length (set through natac_batch_set_track), every write_zero / keep_runs_before_nan combination.
Checked per case: text == natac_write_bedgraph's bytes; the BGZF members inflate to that text and equal natac_bgzf_lines_host byte for
byte; the .tbi from the device's records == the .tbi natac_tabix_index builds from the file.
usage: python tests/fuzz/fuzz_writer.py [rounds] [seed]   (FUZZ_SECONDS bounds the run)"""
import gzip
import io
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from nucleoatac_amd import _lib as L                                   # noqa: E402
from nucleoatac_amd.device import Context                              # noqa: E402
from nucleoatac_amd.packing import PackedChunks                        # noqa: E402
from nucleoatac_amd.writer import BGZF_EOF, TbiBuilder, bgzf_lines_host, tabix_index, write_bedgraph   # noqa: E402


def random_track(rng, n):
    """values with structure: runs, zeros, NaN stretches, magnitudes from 1e-30 to 1e11, integers, repeated neighbours"""
    kind = rng.integers(0, 5)
    if kind == 0:
        v = rng.random(n)
    elif kind == 1:
        v = rng.normal(0, 1, n) * 10.0 ** rng.integers(-30, 11, n)
    elif kind == 2:
        v = rng.poisson(0.4, n).astype(float)
    elif kind == 3:
        v = np.round(rng.random(n), int(rng.integers(0, 4)))
    else:
        v = np.repeat(rng.normal(0, 3, n // 7 + 1), 7)[:n]
    for _ in range(int(rng.integers(0, 6))):                            # runs of one value
        a = int(rng.integers(0, n))
        v[a:a + int(rng.integers(1, 400))] = rng.choice([0.0, 1.0, 0.25, -2.5, 1e-7, 123456.0])
    for _ in range(int(rng.integers(0, 5))):                            # NaN stretches
        a = int(rng.integers(0, n))
        v[a:a + int(rng.integers(1, 300))] = np.nan
    if rng.random() < 0.2:
        v[:int(rng.integers(1, 50))] = np.nan
    if rng.random() < 0.2:
        v[-int(rng.integers(1, 50)):] = 0.0
    return v


def main():
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def run(rounds, seed):
    rng = np.random.default_rng(seed)
    t0 = time.time()
    lines = 0
    done = 0
    tmp = tempfile.mkdtemp(prefix="natac_fuzz_writer_")
    with Context(0) as ctx:
        for r in range(rounds):
            nc = int(rng.integers(1, 60))
            lens = rng.integers(1, 6000, nc)
            if rng.random() < 0.3:
                lens[rng.integers(0, nc)] = int(rng.integers(20000, 90000))       # a chunk longer than a BGZF member's text
            names = ["chr%d" % i for i in range(1, 4)] + ["scaffold_%d_random_%s" % (i, "x" * int(rng.integers(0, 30))) for i in range(2)]
            cid = np.sort(rng.integers(0, len(names), nc))
            chroms = [names[i] for i in cid]
            starts = np.zeros(nc, dtype=np.int64)
            pos = int(rng.integers(0, 10 ** int(rng.integers(1, 9))))
            for k in range(nc):
                if k and cid[k] != cid[k - 1]:
                    pos = int(rng.integers(0, 5000))
                starts[k] = pos
                pos += int(lens[k]) + int(rng.integers(0, 3000))
            off = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
            pk = PackedChunks(starts, lens, np.zeros(nc + 1, dtype=np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64),
                              np.concatenate(([0], np.cumsum(lens + 493))), np.zeros(int((lens + 493).sum())))
            b = ctx.upload(pk)
            try:
                vals = random_track(rng, int(off[-1]))
                b.set_track(L.T_NORM, vals)
                for wz, keep in ((True, False), (False, False), (True, True), (False, True)):
                    p = os.path.join(tmp, "n.bedgraph")
                    write_bedgraph(p, chroms, starts, off, vals, compress=0, write_zero=wz, keep_runs_before_nan=keep)
                    want = open(p, "rb").read()
                    text, ti = b.format_track(L.T_NORM, chroms, starts, write_zero=wz, keep_runs_before_nan=keep, compress=False)
                    assert ti["hard"] == 0
                    assert text.tobytes() == want, ("text", r, wz, keep)
                    z, zi = b.format_track(L.T_NORM, chroms, starts, write_zero=wz, keep_runs_before_nan=keep, compress=True)
                    z = z.tobytes()
                    assert zi["text_bytes"] == len(want) and zi["lines"] == want.count(b"\n")
                    if want:
                        assert gzip.GzipFile(fileobj=io.BytesIO(z + BGZF_EOF)).read() == want, ("inflate", r, wz, keep)
                        assert z == bgzf_lines_host(want), ("members", r, wz, keep)
                        path = os.path.join(tmp, "f.bedgraph.gz")
                        open(path, "wb").write(z + BGZF_EOF)
                        tb = TbiBuilder()
                        tb.push(zi["index"], 0)
                        n_dev = tb.write(path + ".dev.tbi")
                        n_file = tabix_index(path)
                        assert n_dev == n_file and open(path + ".dev.tbi", "rb").read() == open(path + ".tbi", "rb").read(), ("index", r, wz, keep)
                    else:
                        assert len(z) == 0
                    lines += zi["lines"]
            finally:
                b.free()
            done = r + 1
            if time.time() - t0 > float(os.environ.get("FUZZ_SECONDS", "1e9")):
                break
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    print("writer fuzz ok: %d rounds, %d lines, %.0f s" % (done, lines, time.time() - t0))
    return done, lines


if __name__ == "__main__":
    main()
