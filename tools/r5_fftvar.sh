#!/bin/bash
# FFT harness: workgroup shapes, several repetitions interleaved
for rep in 1 2 3 4 5; do for v in 0 20 40; do tools/mb_fft_cur 20000 2120 $v skip | grep -E "^FFT"; done; done
