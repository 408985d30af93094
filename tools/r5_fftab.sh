#!/bin/bash
# A/B of FFT harness binaries: bash tools/r5_fftab.sh "s0 s1" [reps]
for rep in $(seq 1 ${2:-3}); do for b in $1; do echo -n "$b  "; tools/mb_fft_$b 20000 2120 0 skip | grep -E "^FFT"; done; done
for b in $1; do echo -n "$b  "; tools/mb_fft_$b 4000 2120 0 | grep -E "max rel"; done
