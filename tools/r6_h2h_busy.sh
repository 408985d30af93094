#!/bin/bash
# round 6: is the bedGraph.gz host-to-host leg GPU-bound?  union of the kernels' intervals (all streams) over the leg's wall time, and the same for D2H copies
R=$PWD; O=$R/gpurun_out/r6/h2h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/h3; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/h3 -o h --output-format csv -- python $R/bench.py --no-cpu-baseline --cli-chunks 0 --steps 3 --warmup 1 > $O/bench_busy.log 2>&1
python3 - <<'PY' > $O/busy.txt
import csv, glob
def union(iv):
    iv.sort(); tot=0; cs,ce=iv[0]
    for s,e in iv[1:]:
        if s>ce: tot+=ce-cs; cs,ce=s,e
        else: ce=max(ce,e)
    return tot+ce-cs
k=[]; tz=[]
for f in glob.glob("/tmp/h3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"]); k.append((s,e,r["Kernel_Name"]))
c=[]
for f in glob.glob("/tmp/h3/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        c.append((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r.get("Direction",""),int(r.get("Bytes",0) or 0) if "Bytes" in r else 0))
# the text leg = from the first tz_ kernel to the last one
t0=min(s for s,e,n in k if "tz_" in n); t1=max(e for s,e,n in k if "tz_" in n)
kk=[(s,e) for s,e,n in k if e>t0 and s<t1]
print("text leg window %.1f ms (incl. its untimed warm pass)"%((t1-t0)/1e6))
print("kernels busy (union over all streams) %.1f ms = %.1f %%; sum of durations %.1f ms"%(union(kk)/1e6, 100*union(kk)/(t1-t0), sum(e-s for s,e in kk)/1e6))
cc=[(s,e) for s,e,d,b in c if e>t0 and s<t1]
if cc: print("copies busy (union) %.1f ms = %.1f %%; sum %.1f ms; n=%d"%(union(cc)/1e6,100*union(cc)/(t1-t0),sum(e-s for s,e in cc)/1e6,len(cc)))
both=kk+cc
print("kernels OR copies busy %.1f %%"%(100*union(both)/(t1-t0)))
PY
cat $O/busy.txt; grep '^{' $O/bench_busy.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_to_host']; print('text', h['as_bedgraph_gz']['host_to_host_mbp_s'], h['as_bedgraph_gz']['seconds'])"
