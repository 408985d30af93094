#!/bin/bash
# A/B/C: in-tree library, tools/libnatac_exp1.so, tools/libnatac_exp.so on a 20 k-chunk slice
R=$PWD
B="--chunks ${1:-20000} --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0"
for rep in 1 2; do for b in base exp1 exp; do
  L=""; [ $b = exp ] && L=$R/tools/libnatac_exp.so; [ $b = exp1 ] && L=$R/tools/libnatac_exp1.so
  NATAC_LIB=$L timeout 600 python bench.py $B 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['ms_per_step'], d['kernels_ms_per_step'])"
done; done
