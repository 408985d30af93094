#!/bin/bash
# round 6: per-kernel times of the device track writer (tools/prof_textz.py: 20 k chunks, three tracks)
R=$PWD; O=$R/gpurun_out/r6/textz; mkdir -p $O
python tools/prof_textz.py 2>&1 | grep "^track" | sed -e "s/'index'.*}//" | cut -c1-200 > $O/tracks_${1:-base}.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tz; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tz -o tz --output-format csv -- python $R/tools/prof_textz.py > /tmp/tz.log 2>&1
python3 - <<'PY' > $O/kernels_${1:-base}.txt
import csv, glob
rows=[]
for f in glob.glob("/tmp/tz/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "tz_" in r["Name"] or "scan" in r["Name"] or "natac_text" in r["Name"]:
            rows.append((float(r["TotalDurationNs"]), r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"])))
for t,n,c,a in sorted(rows, reverse=True): print("  %-44s calls %4s  avg %8.3f ms  total %8.3f ms" % (n, c, a/1e6, t/1e6))
print("sum tz ms per track (3 tracks): %.2f" % (sum(r[0] for r in rows)/3e6))
PY
cat $O/tracks_${1:-base}.txt $O/kernels_${1:-base}.txt
