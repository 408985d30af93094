#!/bin/bash
# round 5: epilogue inputs where they were (p0) / requested before the inverse transform (p1) / + twiddles requested before the window staging (p2)
O=gpurun_out/r5/ext6; mkdir -p $O
for rep in 1 2 3 4 5; do for b in mb_fft_p0 mb_fft_p1 mb_fft_p2; do echo "== $b variant=1"; timeout 120 tools/$b 20000 2120 1 x | grep -E "^FFT"; done; done > $O/harness2.txt 2>&1
cat $O/harness2.txt
