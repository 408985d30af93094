#!/bin/bash
# round 5, experiment 4 (GPU box): pruned short tile A/B in the FFT harness; library of the last commit against the working tree on the
# default step; the tests that pin the touched kernels
R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O
{
for rep in 1 2 3; do for b in p0 p1; do echo "== $b"; tools/mb_fft_$b 20000 2120 0 skip | grep -E "^FFT"; done; done
echo "== accuracy p0"; tools/mb_fft_p0 4000 2120 0 | grep -E "max rel"
echo "== accuracy p1"; tools/mb_fft_p1 4000 2120 0 | grep -E "max rel"
echo "== L = 10120"; for b in p0 p1; do tools/mb_fft_$b 4000 10120 0 skip | grep -E "^FFT"; done
} > $O/exp4.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_generic_params.py tests/test_gpu_properties.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q > $O/pytest_gpu4.log 2>&1; echo "pytest rc=$?" >> $O/exp4.txt
B="--steps 20 --warmup 3 --no-cpu-baseline --no-h2h --cli-chunks 0"
for rep in 1 2; do for b in head new; do
  L=""; [ $b = head ] && L=$R/tools/libnatac_head.so
  NATAC_LIB=$L timeout 600 python bench.py $B > $O/bench4_$b.$rep.log 2>&1
  grep '^{' $O/bench4_$b.$rep.log > $O/bench4_$b.$rep.json
  python3 -c "import json; d=json.load(open('$O/bench4_$b.$rep.json')); print('$b', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/exp4.txt
done; done
tail -5 $O/pytest_gpu4.log; cat $O/exp4.txt
