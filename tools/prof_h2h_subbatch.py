#!/usr/bin/env python3
"""wall-clock of every call of ONE sub-batch of bench.py's bedGraph.gz host-to-host leg (2,500 chunks of configs[2], five tracks) as an
executor worker issues them, one context, nothing else on the GPU: the sub-batch's own latency chain.  With NATAC_WRITER_DEBUG=1 the
library prints the phases inside natac_batch_format_track.   python tools/prof_h2h_subbatch.py [chunks]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
    import bench
    from nucleoatac_amd import _lib as L
    from nucleoatac_amd.device import Context, pinned_copy, pinned_empty
    from nucleoatac_amd.executor import Stages
    from nucleoatac_amd.packing import PackedChunks
    from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution
    par = np.load(os.path.join(ROOT, "tests", "golden", "params_example.npz"))
    s = make_synthetic_chunks(nc, 2120, 500, seed=1)
    pk = PackedChunks(chunk_start=s.chunk_start, chunk_len=s.chunk_len, frag_off=s.frag_off, frag_lpos=pinned_copy(s.frag_lpos),
                      frag_ilen=pinned_copy(s.frag_ilen), bias_off=s.bias_off, bias_log=pinned_copy(s.bias_log),
                      chroms=["chr%d" % (1 + k % 22) for k in range(nc)])
    nucp, nfrp = synth_occ_distributions(251)
    ctx = Context(0)
    ctx.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"]))
    ctx.set_sizes(synth_size_distribution(251))
    ctx.set_occ_model(nucp, nfrp, step=5, flank=60)
    tracks = [getattr(L, t) for t in bench.H2H_TRACKS]
    st = Stages(nuc_sd=10, occ=True, ins=(0, 2000), peaks=dict(min_signal=0, sep=25, boundary=60, order=12), text_tracks=tracks)
    bufs = {}
    for rep in range(4):
        t = [time.perf_counter()]
        names = []

        def mark(n):
            t.append(time.perf_counter())
            names.append(n)
        b = ctx.upload(pk); mark("upload")
        n = st.run(b); ctx.sync(); mark("stages")
        for tr in tracks:
            def out(nb, tr=tr):
                if tr not in bufs or bufs[tr].size < nb:
                    bufs[tr] = pinned_empty(int(nb * 1.2) + 16, np.uint8)
                return bufs[tr][:nb]
            z, info = b.format_track(tr, pk.chroms, pk.chunk_start, compress=True, out=out)
            mark("format %d (%d MB)" % (tr, len(z) >> 20))
        pkd = b.download_peaks(n); mark("peaks")
        b.status(); mark("status")
        b.free(); mark("free")
        print("rep %d  %.1f Mbp  total %.1f ms: " % (rep, pk.total_bp / 1e6, (t[-1] - t[0]) * 1e3) +
              "  ".join("%s %.2f" % (n_, (b_ - a_) * 1e3) for n_, a_, b_ in zip(names, t[:-1], t[1:])))
    ctx.close()


if __name__ == "__main__":
    main()
