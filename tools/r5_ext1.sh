#!/bin/bash
# round 5, extended FFT tiles: harness A/B (plain tiles vs extended tiles + edge pass) on several chunk lengths, then the GPU tests that pin the background
O=gpurun_out/r5/ext1; mkdir -p $O
for L in 2120 2120 848 4100; do
  n=$((42400000 / L))
  for v in 0 1; do echo "== L=$L variant=$v"; timeout 120 tools/mb_fft $n $L $v | grep -E "^FFT|transforms|max rel|^direct"; done
done > $O/harness.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_golden.py tests/test_gpu_long_chunks.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/harness.txt
tail -5 $O/pytest.log >> $O/harness.txt
for e in 1 0; do NATAC_BG_EXT=$e timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0 2>/dev/null | grep '^{' > $O/bench_ext$e.json; python3 -c "import json; d=json.load(open('$O/bench_ext$e.json')); print('ext=$e', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/harness.txt; done
cat $O/harness.txt
