#!/bin/bash
# writer A/B: tools/libnatac_base.so (last commit) against the in-tree library; then the writer tests and the writer fuzzer
R=$PWD
for rep in 1 2; do for b in base new; do
  L=""; [ $b = base ] && L=$R/tools/libnatac_base.so
  echo "== $b"; NATAC_LIB=$L python tools/prof_textz.py 2>&1 | grep "^track" | sed -e "s/'index'.*}//" | cut -c1-200
done; done
cd /tmp && export TMPDIR=/tmp
for b in base new; do
  L=""; [ $b = base ] && L=$R/tools/libnatac_base.so
  NATAC_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tz_$b -o tz --output-format csv -- python $R/tools/prof_textz.py > /tmp/tz_$b.log 2>&1
  echo "== kernel stats $b"; python3 - /tmp/tz_$b <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tz_" in r["Name"] or "scan" in r["Name"]:
            print("  %-40s calls %4s  avg %8.3f ms  total %8.3f ms" % (r["Name"].split("(")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["TotalDurationNs"]) / 1e6))
PY
done
cd $R
timeout 900 python -m pytest tests/test_gpu_textz.py tests/test_gpu_long_chunks.py tests/test_gpu_resident_occ.py -x -q 2>&1 | tail -4
FUZZ_SECONDS=120 timeout 300 python tests/fuzz/fuzz_writer.py 100000 3 2>&1 | tail -2
