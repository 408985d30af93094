#!/bin/bash
# round 6: the pair loop as it was (s0) against the skewed pair loop (s1): same box, alternating, extended tiles (variant 1) and plain tiles (variant 0)
O=gpurun_out/r6/skew1; mkdir -p $O
{
for b in mb_fft_s0 mb_fft_s1; do echo "== $b variant=1 (accuracy vs direct + output hash)"; timeout 300 tools/$b 20000 2120 1 | grep -E "^FFT|max rel|fnv"; done
for b in mb_fft_s0 mb_fft_s1; do echo "== $b variant=0 (accuracy vs direct + output hash)"; timeout 300 tools/$b 20000 2120 0 | grep -E "^FFT|max rel|fnv"; done
for rep in 1 2 3 4 5; do for b in mb_fft_s0 mb_fft_s1; do echo "== $b variant=1"; timeout 120 tools/$b 20000 2120 1 x | grep -E "^FFT"; done; done
for rep in 1 2 3; do for b in mb_fft_s0 mb_fft_s1; do echo "== $b variant=0"; timeout 120 tools/$b 20000 2120 0 x | grep -E "^FFT"; done; done
} > $O/harness.txt 2>&1
cat $O/harness.txt
