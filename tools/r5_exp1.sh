#!/bin/bash
# round 5, experiment 1 (GPU box, repo root): dft8 h-fold A/B, multi-wave workgroups with / without a per-pair barrier, TCP hit rate of the
# template-spectrum loads; then the GPU test-suite and the driver-style bench line with both builds of the library
R=$PWD; O=$R/gpurun_out/r5; mkdir -p $O
{
for rep in 1 2 3; do for b in h0 h1; do echo "== $b"; tools/mb_fft_$b 20000 2120 0 skip | grep -E "^FFT"; done; done
echo "== accuracy h0"; tools/mb_fft_h0 4000 2120 0 | grep -E "max rel"
echo "== accuracy h1"; tools/mb_fft_h1 4000 2120 0 | grep -E "max rel"
for v in 20 21 40 41 80 81 0; do tools/mb_fft_h1 20000 2120 $v skip | grep -E "^FFT"; done
for v in 41 81; do echo "== accuracy variant $v"; tools/mb_fft_h1 4000 2120 $v | grep -E "max rel"; done
} > $O/exp1_fft.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -E -o "(TCP|TCC)_[A-Z0-9_]+(_sum)?" | sort -u > $O/counters_tcp_tcc.txt
for v in 0 81; do
  timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --kernel-trace -d /tmp/tcp$v -o p --output-format csv -- $R/tools/mb_fft_h1 20000 2120 $v skip > /tmp/tcp$v.log 2>&1
  echo "variant $v rc=$?" >> $O/exp1_tcp.txt
  python3 - /tmp/tcp$v >> $O/exp1_tcp.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        n[(k, r["Counter_Name"])] += 1
for k, d in acc.items():
    print(k, {c: (v / max(1, n[(k, c)])) for c, v in d.items()})
PY
done
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/exp1_fft.txt
for b in h0 h1; do
  L=""; [ $b = h0 ] && L=$R/tools/libnatac_h0.so
  NATAC_LIB=$L timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-h2h --cli-chunks 0 > $O/bench_$b.log 2>&1
  grep '^{' $O/bench_$b.log > $O/bench_$b.json
  python3 -c "import json,sys; d=json.load(open('$O/bench_$b.json')); print('$b', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/exp1_fft.txt
done
cat $O/exp1_fft.txt $O/exp1_tcp.txt; tail -3 $O/pytest_gpu.log
