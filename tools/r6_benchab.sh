#!/bin/bash
# round 6: the configs[2] step with A/B builds of the library on ONE box, alternating (NATAC_LIB): per-kernel ms of the step
O=$PWD/gpurun_out/r6/benchab; mkdir -p $O; : > $O/ab.txt
for rep in 1 2 3; do for l in $LIBS; do
  NATAC_LIB=$PWD/tools/ab/$l.so timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --cli-chunks 0 2>&1 | grep '^{' > /tmp/b.json
  python - $l <<'PY' >> $O/ab.txt
import json,sys
d=json.load(open("/tmp/b.json")); k=d["kernels_ms_per_step"]
print(sys.argv[1], "step %.2f ms  %.1f Mbp/s  background %.2f  cand %.2f occ %.2f gather %.2f clock_bg %.3f"%(d["ms_per_step"], d["value"], k["background"], k["candidates"], k["occ_mle"], k["frag_gather"], d["roofline"]["clock_ghz"]["background"]))
PY
done; done
cat $O/ab.txt
