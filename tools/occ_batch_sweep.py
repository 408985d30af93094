#!/usr/bin/env python3
"""`nucleoatac occ` end to end over (sub-batch size, contexts): N windows of length L with F fragments as .npz inputs, then one run per
setting.   python tools/occ_batch_sweep.py N L F  bp:9000000,ctx:3  bp:4500000,ctx:3  chunks:4096,ctx:3 ..."""
import contextlib
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nucleoatac_amd import pipeline  # noqa: E402
from nucleoatac_amd.nucleoatac import run_occ as ro  # noqa: E402
from nucleoatac_amd.nucleoatac.cli import main as cli_main  # noqa: E402
from nucleoatac_amd.synth import write_cli_dataset  # noqa: E402

n, L, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = tempfile.mkdtemp(prefix="natac_occ_sweep_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    bed, bam, fa = write_cli_dataset(d, n, L, F, seed=0)
    for i, spec in enumerate(sys.argv[4:]):
        kv = dict(x.split(":") for x in spec.split(","))
        ro.SUB_BATCH_BP = int(kv.get("bp", pipeline.SUB_BATCH_BP))
        ro.BATCH_CHUNKS = int(kv.get("chunks", 4096))
        if "chunks" in kv and "bp" not in kv:
            ro.SUB_BATCH_BP = 1 << 62
        ro.N_CONTEXTS = int(kv.get("ctx", 3))
        out = os.path.join(d, "o%d" % i)
        with contextlib.redirect_stdout(sys.stderr):
            t0 = time.perf_counter()
            cli_main(["occ", "--bed", bed, "--bam", bam, "--fasta", fa, "--out", out])
            dt = time.perf_counter() - t0
        ph = ro.LAST_TIMINGS
        print(json.dumps(dict(setting=spec, occ_seconds=round(dt, 2), mbp_s=round(n * L / dt / 1e6, 1), pipeline_wall=ph.get("pipeline_wall"),
                              first_gap_last=ph.get("results_first_median_gap_last"), writer=ph.get("writer_inside_pipeline"),
                              pack=ph.get("pack_inside_pipeline"))))
        for f in os.listdir(d):
            if f.startswith("o%d." % i):
                os.remove(os.path.join(d, f))
finally:
    shutil.rmtree(d, ignore_errors=True)
