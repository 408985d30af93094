#!/bin/bash
# round 6: where the bedGraph.gz host-to-host leg spends the GPU: kernel-trace statistics of a bench run whose only long leg is that one
R=$PWD; O=$R/gpurun_out/r6/h2h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/h2; timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/h2 -o h --output-format csv -- python $R/bench.py --no-cpu-baseline --cli-chunks 0 --steps 3 --warmup 1 > $O/bench.log 2>&1
python3 - <<'PY' > $O/kernels.txt
import csv, glob, json
rows=[]
for f in glob.glob("/tmp/h2/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"]), r["Name"].split("(")[0][-50:], int(r["Calls"]), float(r["AverageNs"])))
tz=sum(t for t,n,c,a in rows if "tz_" in n or "textz" in n)
print("all kernels %.1f ms, tz_* %.1f ms" % (sum(r[0] for r in rows)/1e6, tz/1e6))
for t,n,c,a in sorted(rows, reverse=True)[:28]: print("  %-50s calls %5d  avg %8.3f ms  total %9.3f ms" % (n, c, a/1e6, t/1e6))
l=[x for x in open("/root/repo/gpurun_out/r6/h2h/bench.log") if x.startswith("{")][-1]
d=json.loads(l); h=d["host_to_host"]
print("step", d["ms_per_step"], "h2h f64", h["host_to_host_mbp_s"], h["seconds"], "text", h["as_bedgraph_gz"]["host_to_host_mbp_s"], h["as_bedgraph_gz"]["seconds"], "steps", h["steps"])
PY
cat $O/kernels.txt
