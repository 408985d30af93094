#!/usr/bin/env python3
"""where the six worker threads of bench.py's bedGraph.gz host-to-host leg spend their wall time: every ctypes call of the library is
wrapped with a timer (the GIL is released inside), so thread time = sum of call times + Python in between (GIL held or waited for).
python tools/prof_h2h_threads.py [steps]"""
import collections
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    import bench
    from nucleoatac_amd import _lib as L
    from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions, synth_size_distribution
    lib = L.load()
    acc = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    lock = threading.Lock()

    class Wrapped(object):
        def __init__(self, lib):
            object.__setattr__(self, "_lib", lib)
            object.__setattr__(self, "_cache", {})

        def __getattr__(self, name):
            c = self._cache.get(name)
            if c is None:
                fn = getattr(self._lib, name)

                def call(*a, _fn=fn, _n=name):
                    t0 = time.perf_counter()
                    r = _fn(*a)
                    dt = time.perf_counter() - t0
                    with lock:
                        acc[_n] += dt
                        cnt[_n] += 1
                    return r
                c = self._cache[name] = call
            return c
    import nucleoatac_amd.device as D
    w = Wrapped(lib)
    orig_load = L.load
    L.load = lambda: w
    D.L.load = L.load
    par = np.load(os.path.join(ROOT, "tests", "golden", "params_example.npz"))
    pk = make_synthetic_chunks(100000, 2120, 500, seed=0)
    nucp, nfrp = synth_occ_distributions(251)
    t0 = time.perf_counter()
    h = bench.host_to_host(pk, 0, par, synth_size_distribution(251), nucp, nfrp, steps, 2500, 6, as_text=True)
    wall = time.perf_counter() - t0
    print("text leg %.1f Mbp/s, timed part %.3f s (whole call %.2f s incl. the untimed pass)" % (h["host_to_host_mbp_s"], h["seconds"], wall))
    tot = sum(acc.values())
    print("sum of library-call time over all threads %.2f s = %.2f threads busy inside the library on average" % (tot, tot / wall))
    for k, v in sorted(acc.items(), key=lambda x: -x[1])[:14]:
        print("  %-40s calls %6d  total %7.3f s  avg %7.3f ms" % (k, cnt[k], v, 1e3 * v / cnt[k]))


if __name__ == "__main__":
    main()
