"""profile of the device track writer on a 20,000-chunk batch (run under rocprofv3 --kernel-trace --stats)"""
import sys, time, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nucleoatac_amd import _lib as L
from nucleoatac_amd.device import Context
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution
par = np.load(os.path.join(ROOT, "tests", "golden", "params_example.npz"))
c = Context(0); c.set_vmat(par["vmat"], int(par["vlower"]), int(par["vupper"])); c.set_sizes(synth_size_distribution(251))
nucp, nfrp = synth_occ_distributions(251); c.set_occ_model(nucp, nfrp, step=5, flank=60)
pk = make_synthetic_chunks(20000, 2120, 500, seed=3)
chroms = ["chr%d" % (1 + k // 5000) for k in range(pk.n_chunks)]
b = c.upload(pk); b.run_occ(); b.run_nuc(10)
from nucleoatac_amd.device import pinned_empty
buf = pinned_empty(1 << 30, np.uint8)
for t in (L.T_OCC, L.T_NORM, L.T_OCC):
    t0 = time.perf_counter(); z, info = b.format_track(t, chroms, pk.chunk_start, compress=True, out=lambda n: buf[:n]); dt = time.perf_counter() - t0
    print("track", t, info, "%.3f s  %.0f Mbp/s" % (dt, pk.total_bp / dt / 1e6))
