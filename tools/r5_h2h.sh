#!/bin/bash
# host-to-host rates (float64 tracks / .bedgraph.gz bytes) of tools/libnatac_base.so against the in-tree library, same box
R=$PWD
for rep in 1 2; do for b in base new; do
  L=""; [ $b = base ] && L=$R/tools/libnatac_base.so
  NATAC_LIB=$L timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --cli-chunks 0 2>/dev/null | grep '^{' | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); h=d['host_to_host']; print('$b', d['ms_per_step'], h['host_to_host_mbp_s'], h['as_bedgraph_gz']['host_to_host_mbp_s'], h['as_bedgraph_gz'].get('gb_down_per_step'))"
done; done
