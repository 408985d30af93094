# development: A/B of a compile-time switch of natac_fft_bg.hpp in the stand-alone harness:  bash tools/fft_ab.sh "0 1" NATAC_FFT_EPI_PREFETCH
F="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -pragma-unroll-threshold=1000000"
for k in ${1:-0}; do
  hipcc $F -D${2:-NATAC_FFT_EPI_PREFETCH}=$k tools/test_fft_bg.hip -o /tmp/mb_fft_k$k 2>&1 | grep -E "error" | head
done
for rep in 1 2; do for k in ${1:-0}; do
  echo "== ${2:-NATAC_FFT_EPI_PREFETCH}=$k"; /tmp/mb_fft_k$k 20000 2120 | grep -E "^FFT|max rel"
done; done
