#!/bin/bash
# round 5: harness timing, plain tiles (0) vs extended tiles with the vector-pipe edge pass (1) or the matrix-pipe edge pass (2)
O=gpurun_out/r5/ext3; mkdir -p $O
for rep in 1 2; do for v in 0 1 2; do echo "== variant=$v"; timeout 120 tools/mb_fft 20000 2120 $v | grep -E "^FFT|transforms|^bg|^norm|^bnum|^bcov"; done; done > $O/harness.txt 2>&1
for L in 848 4100; do n=$((42400000 / L)); echo "== L=$L variant=2"; timeout 120 tools/mb_fft $n $L 2 | grep -E "^FFT|transforms|^bg|^norm|^bnum|^bcov"; done >> $O/harness.txt 2>&1
cat $O/harness.txt
timeout 900 python -m pytest tests/test_gpu_properties.py tests/test_gpu_golden.py tests/test_gpu_long_chunks.py -x -q -m gpu 2>&1 | tail -5
for e in 1 0; do NATAC_BG_EDGE_MFMA=$e timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0 2>/dev/null | grep '^{' > $O/bench_mfma$e.json; python3 -c "import json; d=json.load(open('$O/bench_mfma$e.json')); print('mfma=$e', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])"; done
