#!/usr/bin/env python3
"""wall-clock of every call of one `occ` sub-batch as the executor's worker issues them (upload, stages, device text of the three
tracks, peaks, adoption), one context, repeated: where a sub-batch's latency goes.   python tools/prof_subbatch.py [chunks] [L] [F]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    L_ = int(sys.argv[2]) if len(sys.argv) > 2 else 10120
    F = int(sys.argv[3]) if len(sys.argv) > 3 else 667
    from nucleoatac_amd import _lib as L
    from nucleoatac_amd.device import Context, TrackStore, pinned_empty
    from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions
    pk = make_synthetic_chunks(nc, L_, F, seed=1)
    pk.chroms = ["chr%d" % (1 + k % 4) for k in range(nc)]
    nucp, nfrp = synth_occ_distributions(251)
    ctx = Context(0)
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    store = TrackStore()
    bufs = {}
    for rep in range(3):
        t = [time.perf_counter()]
        names = []

        def mark(n):
            ctx.sync()
            t.append(time.perf_counter())
            names.append(n)
        b = ctx.upload(pk); mark("upload")
        b.run_occ(); mark("run_occ")
        for tr in (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER):
            def out(nb, tr=tr):
                if tr not in bufs or bufs[tr].size < nb:
                    bufs[tr] = pinned_empty(int(nb * 1.2) + 16, np.uint8)      # first repetition only (the executor's slots do the same)
                return bufs[tr][:nb]
            z, info = b.format_track(tr, pk.chroms, pk.chunk_start, compress=True, out=out)
            mark("format_track %d (%d MB)" % (tr, len(z) >> 20))
        pkd = b.run_occ_peaks(min_occ=0.1, sep=120); mark("occ_peaks")
        st = b.status(); mark("status")
        seg = store.adopt(b, (L.T_OCC, L.T_OCC_LOWER, L.T_OCC_UPPER)); mark("adopt")
        b.free(); mark("free")
        print("rep %d  %.1f Mbp  total %.1f ms: " % (rep, pk.total_bp / 1e6, (t[-1] - t[0]) * 1e3) +
              "  ".join("%s %.1f" % (n, (b_ - a_) * 1e3) for n, a_, b_ in zip(names, t[:-1], t[1:])))
    store.close()
    ctx.close()


if __name__ == "__main__":
    main()
