#!/bin/bash
R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O
{
for rep in 1 2 3 4; do for b in base c1; do echo "== $b"; tools/mb_fft_$b 20000 2120 0 skip | grep -E "^FFT"; done; done
echo "== accuracy c1"; tools/mb_fft_c1 4000 2120 0 | grep -E "max rel"
} > $O/exp6.txt 2>&1
cat $O/exp6.txt
