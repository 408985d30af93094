#!/bin/bash
# host-to-host rates (float64 arrays / bedGraph.gz bytes) of bench.py over contexts x sub-batch size.   usage: bash tools/h2h_sweep.sh
mkdir -p gpurun_out/h2h
for cfg in "6 2500" "8 2500" "12 2500" "6 5000" "8 5000" "4 10000"; do
  set -- $cfg
  timeout 600 python bench.py --no-cpu-baseline --cli-chunks 0 --h2h-threads $1 --h2h-sub $2 > gpurun_out/h2h/t$1_s$2.json 2> gpurun_out/h2h/t$1_s$2.err
  python - gpurun_out/h2h/t$1_s$2.json $1 $2 <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    h = d["host_to_host"]
    print("contexts %s sub %s: f64 %.0f Mbp/s (%.1f GB/s down)   bedgraph.gz %.0f Mbp/s (%.1f GB/s down, %.2f GB per step)" % (
        sys.argv[2], sys.argv[3], h["host_to_host_mbp_s"], h["pcie_gbs_down"], h["as_bedgraph_gz"]["host_to_host_mbp_s"],
        h["as_bedgraph_gz"]["pcie_gbs_down"], h["as_bedgraph_gz"]["gb_down_per_step"]))
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
