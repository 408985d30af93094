for rep in 1 2; do for cfg in "6 2500" "8 2500" "10 2500" "8 5000" "10 5000"; do set -- $cfg
  python bench.py --no-cpu-baseline --cli-chunks 0 --steps 4 --warmup 2 --h2h-threads $1 --h2h-sub $2 2>&1 | grep "^{" > /tmp/b.json
  python - $1 $2 <<PY
import json,sys
d=json.load(open("/tmp/b.json")); h=d["host_to_host"]
print("contexts", sys.argv[1], "sub", sys.argv[2], "h2h f64", h["host_to_host_mbp_s"], "text", h["as_bedgraph_gz"]["host_to_host_mbp_s"])
PY
done; done
