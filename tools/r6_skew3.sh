#!/bin/bash
# round 6: what the skewed pair loop waits for -- ablations of the same loop (wrong results, same arithmetic): NATAC_SKEW_ABL bits
# 1 = no transpose stores, 2 = no transpose loads, 4 = no template-spectrum loads, 8 = no operand reads; s0 = the loop as it was
O=$PWD/gpurun_out/r6/skew3; mkdir -p $O
T=$PWD/tools
{
for lo in 105 104; do for b in mb_fft_s0 mb_fft_a0; do echo "== $b variant=1 lower=$lo (accuracy vs direct + output hash)"; NATAC_HARNESS_LO=$lo timeout 300 $T/$b 20000 2120 1 | grep -E "^FFT|bg: max rel|fnv"; done; done
for rep in 1 2 3; do for b in mb_fft_s0 mb_fft_a0 mb_fft_a1 mb_fft_a2 mb_fft_a3 mb_fft_a4 mb_fft_a7 mb_fft_a8 mb_fft_a15; do echo "== $b variant=1"; timeout 120 $T/$b 20000 2120 1 x | grep -E "^FFT"; done; done
} > $O/harness.txt 2>&1
cat $O/harness.txt
