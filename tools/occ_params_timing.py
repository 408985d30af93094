"""time of the occupancy stage (natac_run_occ: block sums + decision + smoothing + fill) for non-default --step / --flank next to
the defaults, on N chunks of the configs[2] workload (no 5x cliff off the defaults).
usage (GPU box): python tools/occ_params_timing.py [n_chunks] > gpurun_out/occ_params.json"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nucleoatac_amd.device import Context  # noqa: E402
from nucleoatac_amd.synth import make_synthetic_chunks, synth_occ_distributions  # noqa: E402


def main():
    nc = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    pk = make_synthetic_chunks(nc, 2120, 500, seed=0)
    nucp, nfrp = synth_occ_distributions(251)
    rows = []
    for step, flank, general in ((5, 60, False), (5, 60, True), (3, 60, False), (3, 60, True), (5, 61, False), (5, 61, True), (7, 60, False),
                                 (9, 63, False), (1, 60, False)):
        if general:
            os.environ["NATAC_OCC_GENERAL"] = "1"
        try:
            with Context(0) as c:
                c.set_occ_model(nucp, nfrp, step=step, flank=flank)
                b = c.upload(pk)
                b.run_occ()
                c.sync()
                t0 = time.perf_counter()
                for _ in range(3):
                    b.run_occ()
                c.sync()
                ms = (time.perf_counter() - t0) / 3 * 1e3
                b.free()
        finally:
            os.environ.pop("NATAC_OCC_GENERAL", None)
        grid = nc * len(range((step - 1) // 2, 2120, step))
        rows.append(dict(step=step, flank=flank, kernel="general (natac_occ_mle)" if general else "block sums + decision", ms=round(ms, 3),
                         grid_points=grid, ns_per_grid_point=round(ms * 1e6 / grid, 2)))
        print(rows[-1], file=sys.stderr)
    print(json.dumps(dict(chunks=nc, bp=int(pk.total_bp), rows=rows)))


if __name__ == "__main__":
    main()
