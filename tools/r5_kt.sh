#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the in-tree library against tools/libnatac_exp.so on a 20 k-chunk slice
# usage: bash tools/r5_kt.sh [pattern]
R=$PWD; cd /tmp && export TMPDIR=/tmp
for b in base exp; do
  L=""; [ $b = exp ] && L=$R/tools/libnatac_exp.so
  rm -rf /tmp/kt_$b
  NATAC_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$b -o kt --output-format csv -- python $R/bench.py --chunks 20000 --steps 5 --warmup 1 --no-cpu-baseline --no-h2h --cli-chunks 0 > /tmp/kt_$b.log 2>&1
  echo "== $b"; python3 - /tmp/kt_$b "${1:-natac}" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Name"] and float(r["TotalDurationNs"]) > 2e5:
            print("  %-46s calls %3s  avg %8.3f ms" % (r["Name"].split("(")[0][-46:], r["Calls"], float(r["AverageNs"]) / 1e6))
PY
done
