"""throughput of the native track writer (host cores of the GPU box): Mbp/s for text and BGZF output"""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nucleoatac_amd.writer import write_bedgraph

nc, L = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, 2120
rng = np.random.default_rng(0)
vals = rng.normal(size=nc * L)
off = np.arange(nc + 1) * L
starts = np.arange(nc) * 3120 + 10000
chroms = ["chr%d" % (1 + i * 24 // nc) for i in range(nc)]
d = tempfile.mkdtemp(dir="/tmp")
for comp, thr in ((0, 0), (4, 0), (1, 0), (4, 1)):
    p = os.path.join(d, "t%d.bedgraph" % comp)
    t = time.time()
    nb = write_bedgraph(p, chroms, starts, off, vals, compress=comp, n_threads=thr)
    dt = time.time() - t
    print("compress=%d threads=%s: %.2f s, %.1f Mbp/s, %.1f MB/s out (%d cores)" % (comp, thr or "auto", dt, nc * L / dt / 1e6, nb / dt / 1e6, os.cpu_count()))
    os.remove(p)
