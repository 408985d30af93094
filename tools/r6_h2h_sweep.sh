#!/bin/bash
# round 6: the bedGraph.gz host-to-host leg over (contexts, chunks per sub-batch), three runs each, one box
O=$PWD/gpurun_out/r6/h2h; mkdir -p $O; : > $O/sweep.txt
for rep in 1 2 3; do for cfg in "6 2500" "10 5000" "12 5000" "8 10000" "12 2500"; do set -- $cfg
  python bench.py --no-cpu-baseline --cli-chunks 0 --steps 4 --warmup 2 --h2h-threads $1 --h2h-sub $2 2>&1 | grep "^{" > /tmp/b.json
  python - $1 $2 <<PY >> $O/sweep.txt
import json,sys
d=json.load(open("/tmp/b.json")); h=d["host_to_host"]
print("contexts", sys.argv[1], "sub", sys.argv[2], "h2h f64", h["host_to_host_mbp_s"], "text", h["as_bedgraph_gz"]["host_to_host_mbp_s"], "GB/s down", h["as_bedgraph_gz"]["pcie_gbs_down"])
PY
done; done
sort $O/sweep.txt
