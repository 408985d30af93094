#!/bin/bash
# round 6: the other BASELINE configs at full size on one GPU + long fuzzers (final build)
O=$PWD/gpurun_out/r6/extra; mkdir -p $O
timeout 1500 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline --no-h2h --cli-chunks 0 > $O/bench_cfg4_full.log 2>&1; grep '^{' $O/bench_cfg4_full.log > $O/bench_cfg4_full_300k_tiles_one_gpu.json
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0 > $O/bench_cfg5.log 2>&1; grep '^{' $O/bench_cfg5.log > $O/bench_cfg5.json
python3 - <<'PY' > $O/summary.txt
import json
for n in ("bench_cfg4_full_300k_tiles_one_gpu","bench_cfg5"):
    try:
        d=json.load(open("gpurun_out/r6/extra/%s.json"%n)); print(n, d["value"], d["ms_per_step"], d.get("cov_sweep",{}).get("worst_closed_fp64"), d.get("cov_sweep",{}).get("worst_fp32"))
    except Exception as e: print(n, "failed", e)
PY
FUZZ_SECONDS=900 timeout 1100 python tests/fuzz/fuzz_parity.py 1000000 61 > $O/fuzz_long.log 2>&1; echo "fuzz_parity rc=$?" >> $O/summary.txt
FUZZ_SECONDS=400 timeout 600 python tests/fuzz/fuzz_generic.py 1000000 62 >> $O/fuzz_long.log 2>&1; echo "fuzz_generic rc=$?" >> $O/summary.txt
FUZZ_SECONDS=300 timeout 500 python tests/fuzz/fuzz_writer.py 1000000 63 >> $O/fuzz_long.log 2>&1; echo "fuzz_writer rc=$?" >> $O/summary.txt
FUZZ_SECONDS=200 timeout 400 python tests/fuzz/fuzz_dropins.py 1000000 64 >> $O/fuzz_long.log 2>&1; echo "fuzz_dropins rc=$?" >> $O/summary.txt
cat $O/summary.txt; grep -i "ok" $O/fuzz_long.log | tail -5
