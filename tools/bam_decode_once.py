#!/usr/bin/env python3
"""decode one .bam on the device (natac_bam_open_device) `reps` times; with `make N` first writes a synthetic coordinate-sorted BAM of
N paired-end records (tools/bench_bam.py's generator) to the path.  For rocprofv3 passes over the decoder's kernels.
usage: python tools/bam_decode_once.py PATH [reps] | python tools/bam_decode_once.py PATH make N"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    path = sys.argv[1]
    if len(sys.argv) > 3 and sys.argv[2] == "make":
        from bench_bam import synth_bam
        raw, kept = synth_bam(path, int(sys.argv[3]))
        print("wrote %s: %d bytes compressed, %d inflated, %d kept reads" % (path, os.path.getsize(path), raw, kept))
        return
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from nucleoatac_amd.pyatac.fragments import FragmentStore
    for _ in range(reps):
        t0 = time.perf_counter()
        st = FragmentStore.from_bam(path, device=True)
        dt = time.perf_counter() - t0
        n = sum(len(st.pos[c]) for c in st.pos)
        print("device=%s  kept %d reads  %.3f s  %.0f MB/s compressed" % (FragmentStore.last_bam_on_device, n, dt, os.path.getsize(path) / dt / 1e6))


if __name__ == "__main__":
    main()
