#!/bin/bash
# round 5: the tiling rule on other chunk lengths (harness, plain tiles vs the library's mix), then the GPU tests around the background
O=gpurun_out/r5/ext4; mkdir -p $O
for L in 2000 1200 9900 4100 2120; do n=$((42400000 / L)); for v in 0 1; do echo "== L=$L variant=$v"; timeout 120 tools/mb_fft $n $L $v x | grep -E "^FFT|transforms|^bg"; done; done > $O/harness.txt 2>&1
cat $O/harness.txt
timeout 900 python -m pytest tests/test_gpu_bg_ext.py tests/test_gpu_properties.py tests/test_gpu_golden.py tests/test_gpu_long_chunks.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -5
