#!/bin/bash
# round 5: edge pass as its own kernel (mb_fft) against the edge pass at the end of the transform's own wave (mb_fft_f1: NATAC_FFT_EDGE_FUSED=1)
O=gpurun_out/r5/ext5; mkdir -p $O
for rep in 1 2 3; do for b in mb_fft mb_fft_f1; do for v in 0 1; do echo "== $b variant=$v"; timeout 120 tools/$b 20000 2120 $v | grep -E "^FFT|transforms|^bg|^norm|^bnum|^bcov"; done; done; done > $O/harness.txt 2>&1
for L in 2000 4100; do n=$((42400000 / L)); for b in mb_fft mb_fft_f1; do echo "== $b L=$L variant=1"; timeout 120 tools/$b $n $L 1 | grep -E "^FFT|transforms|^bg|^norm|^bnum|^bcov"; done; done >> $O/harness.txt 2>&1
cat $O/harness.txt
