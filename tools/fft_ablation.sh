#!/bin/bash
# development: what the LDS transposes / template loads / operand reads of natac_background_fft cost (tools/test_fft_bg.hip)
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000"
for abl in 0 1 2 3; do
  hipcc $F -DNATAC_FFT_ABL=$abl tools/test_fft_bg.hip -o /tmp/mb_fft_$abl 2>/dev/null
  echo "== ABL=$abl"; /tmp/mb_fft_$abl 20000 2120 | grep -E "^FFT|max rel"
done
