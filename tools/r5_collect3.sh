#!/bin/bash
# round 5, last build: the background tests, kernel statistics + PMC passes + floor table, then the contract line, a 20-step line and the
# plain-tile line with that summary in place
R=$PWD; O=$R/gpurun_out/r5/final4; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bg_ext.py tests/test_gpu_properties.py tests/test_gpu_golden.py tests/test_gpu_long_chunks.py tests/test_gpu_configs.py tests/test_gpu_generic_params.py -x -q -m gpu > $O/pytest_background.log 2>&1; tail -2 $O/pytest_background.log
bash tools/collect_profiles.sh $O > $O/collect.log 2>&1
python tools/floor_table.py $O/pmc_summary.csv $O/kernel_stats_bench_steps3.csv > $O/floor_table.md
cp $O/pmc_summary.csv profiles/r5/pmc_summary.csv
timeout 500 python bench.py > $O/bench_default.log 2>&1; grep '^{' $O/bench_default.log > $O/bench_default.json
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2h --cli-chunks 0 2>/dev/null | grep '^{' > $O/bench_steps20.json
NATAC_BG_EXT=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2h --cli-chunks 0 2>/dev/null | grep '^{' > $O/bench_steps20_plain_tiles.json
FUZZ_SECONDS=120 timeout 300 python tests/fuzz/fuzz_parity.py 100000 41 2>&1 | tail -1
FUZZ_SECONDS=60 timeout 300 python tests/fuzz/fuzz_generic.py 100000 41 2>&1 | tail -1
python3 -c "
import json
for f in ('bench_default','bench_steps20','bench_steps20_plain_tiles'):
    d=json.load(open('$O/%s.json'%f)); print(f, d['value'], d['ms_per_step'], d['roofline']['traffic_source']['stale'], d['kernels_ms_per_step'])"
cat $O/floor_table.md
