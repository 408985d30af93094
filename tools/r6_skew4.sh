#!/bin/bash
# round 6: s0 = round 5's loop; s0x = the same with 16-byte template-spectrum loads; a0k0 = skewed loop, 8-byte loads; a0 = skewed loop, 16-byte loads; aN = ablations of a0
O=$PWD/gpurun_out/r6/skew4; mkdir -p $O
T=$PWD/tools
{
for lo in 105 104; do for b in mb_fft_s0 mb_fft_a0; do echo "== $b variant=1 lower=$lo (accuracy vs direct + output hash)"; NATAC_HARNESS_LO=$lo timeout 300 $T/$b 20000 2120 1 | grep -E "^FFT|bg: max rel|fnv"; done; done
for rep in 1 2 3 4; do for b in $BINS; do echo "== $b variant=1"; timeout 120 $T/$b 20000 2120 1 x | grep -E "^FFT"; done; done
} > $O/harness.txt 2>&1
cat $O/harness.txt
