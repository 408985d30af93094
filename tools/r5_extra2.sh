#!/bin/bash
# round 5 release checks on the final build (extended tiles): the other workloads on one GPU, long fuzz runs
R=$PWD; O=$R/gpurun_out/r5/extra2; mkdir -p $O
timeout 900 python bench.py --workload cfg5 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_cfg5.log 2>&1; grep '^{' $O/bench_cfg5.log > $O/bench_cfg5.json
timeout 1500 python bench.py --workload cfg4 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_cfg4_full.log 2>&1; grep '^{' $O/bench_cfg4_full.log > $O/bench_cfg4_full_300k_tiles_one_gpu.json
for f in fuzz_parity fuzz_generic; do FUZZ_SECONDS=420 timeout 700 python tests/fuzz/$f.py 1000000 31 >> $O/fuzz_long.log 2>&1; echo "$f rc=$?" >> $O/summary.txt; done
for f in fuzz_writer fuzz_round4 fuzz_dropins; do FUZZ_SECONDS=120 timeout 400 python tests/fuzz/$f.py 1000000 31 >> $O/fuzz_long.log 2>&1; echo "$f rc=$?" >> $O/summary.txt; done
for f in bench_cfg5 bench_cfg4_full_300k_tiles_one_gpu; do python3 -c "import json; d=json.load(open('$O/$f.json')); print('$f', d['value'], d['ms_per_step'], d['config'].get('outputs_recycled'))" >> $O/summary.txt; done
cat $O/summary.txt; grep -i ' ok' $O/fuzz_long.log | tail -6
