#!/bin/bash
# round 5, experiment 3 (GPU box): full GPU suite on the generalised occupancy path + global peak lists; occupancy stage timings off the
# defaults; heavy-tailed workload with / without the launch order
R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu3.log 2>&1; echo "pytest rc=$?" > $O/exp3.txt
timeout 600 python tools/occ_params_timing.py 20000 > $O/occ_params.json 2>> $O/exp3.txt
B="--steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0"
for ord in 1 0; do
for w in "cfg3 --frags-per-chunk 545" "cfg3-heavy"; do
  n=$(echo $w | tr ' ' '_' | tr -d '-')
  NATAC_OCC_ORDER=$ord timeout 600 python bench.py --workload $w $B > $O/bench3_$n.ord$ord.log 2>&1
  grep '^{' $O/bench3_$n.ord$ord.log > $O/bench3_$n.ord$ord.json
  python3 -c "import json; d=json.load(open('$O/bench3_$n.ord$ord.json')); print('order=$ord $w', d['config']['fragments_total'], d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/exp3.txt
done; done
tail -15 $O/pytest_gpu3.log; cat $O/exp3.txt
