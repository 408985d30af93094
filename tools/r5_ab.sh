#!/bin/bash
# A/B of tools/libnatac_exp.so against the in-tree library on a 20 k-chunk slice of the default step (kernel classes, ms per step)
R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O
B="--chunks ${1:-20000} --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0 ${2:-}"
for rep in 1 2; do for b in base exp; do
  L=""; [ $b = exp ] && L=$R/tools/libnatac_exp.so
  NATAC_LIB=$L timeout 600 python bench.py $B 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$b', d['ms_per_step'], d['kernels_ms_per_step'])"
done; done
