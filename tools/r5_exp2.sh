#!/bin/bash
# round 5, experiment 2 (GPU box): the new tests; heavy-tailed workload against the uniform one at the same fragment total, same box
R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_long_chunks.py tests/test_gpu_resident_occ.py -x -q > $O/pytest_new.log 2>&1; echo "pytest rc=$?" > $O/exp2.txt
B="--steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0"
for rep in 1 2; do
for w in "cfg3 --frags-per-chunk 500" "cfg3 --frags-per-chunk 545" "cfg3-heavy"; do
  n=$(echo $w | tr ' ' '_' | tr -d '-')
  timeout 600 python bench.py --workload $w $B > $O/bench_$n.$rep.log 2>&1
  grep '^{' $O/bench_$n.$rep.log > $O/bench_$n.$rep.json
  python3 -c "import json; d=json.load(open('$O/bench_$n.$rep.json')); print('$w', d['config']['fragments_total'], d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/exp2.txt
done; done
tail -15 $O/pytest_new.log; cat $O/exp2.txt
