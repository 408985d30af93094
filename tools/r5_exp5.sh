#!/bin/bash
# round 5, experiment 5 (GPU box): full GPU suite; library of the last commit against the working tree (default step); heavy-tailed
# workload with / without the heavy-first launch against the uniform one at the same fragment total
R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu5.log 2>&1; echo "pytest rc=$?" > $O/exp5.txt
B="--steps 20 --warmup 3 --no-cpu-baseline --no-h2h --cli-chunks 0"
for rep in 1 2; do for b in head new; do
  L=""; [ $b = head ] && L=$R/tools/libnatac_head.so
  NATAC_LIB=$L timeout 600 python bench.py $B > $O/bench5_$b.$rep.log 2>&1
  grep '^{' $O/bench5_$b.$rep.log > $O/bench5_$b.$rep.json
  python3 -c "import json; d=json.load(open('$O/bench5_$b.$rep.json')); print('$b', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/exp5.txt
done; done
B="--steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0"
for w in "cfg3 --frags-per-chunk 545" "cfg3-heavy"; do for ord in 1 0; do
  n=$(echo $w | tr ' ' '_' | tr -d '-')
  NATAC_OCC_ORDER=$ord timeout 600 python bench.py --workload $w $B > $O/bench5_$n.ord$ord.log 2>&1
  grep '^{' $O/bench5_$n.ord$ord.log > $O/bench5_$n.ord$ord.json
  python3 -c "import json; d=json.load(open('$O/bench5_$n.ord$ord.json')); print('heavy-first=$ord $w', d['config']['fragments_total'], d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/exp5.txt
done; done
tail -15 $O/pytest_gpu5.log; cat $O/exp5.txt
