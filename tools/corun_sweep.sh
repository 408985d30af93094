#!/bin/bash
# A/B of the co-scheduled nuc + occ stages (natac_run_nuc_occ) on the configs[2] step: NATAC_CORUN=0 | 1, wave priority of the
# persistent background launch, then a rocprofv3 kernel trace of the co-scheduled run (overlap of the two streams).
# usage (GPU box): bash tools/corun_sweep.sh [steps]      -> gpurun_out/corun/
set -u
STEPS=${1:-10}
OUT=gpurun_out/corun
mkdir -p $OUT
B="python bench.py --steps $STEPS --warmup 2 --no-h2h --cli-chunks 0 --no-cpu-baseline"
for v in "0 0" "1 0" "1 3"; do
  set -- $v
  NATAC_CORUN=$1 NATAC_CORUN_PRIO=$2 timeout 600 $B > $OUT/bench_corun$1_prio$2.json 2> $OUT/bench_corun$1_prio$2.err
  python - $OUT/bench_corun$1_prio$2.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1], "value", d["value"], "ms/step", d["ms_per_step"], d["kernels_ms_per_step"])
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PY
done
cd /tmp && export TMPDIR=/tmp
R=$OLDPWD
NATAC_CORUN=1 timeout 900 rocprofv3 --kernel-trace --stats -d $R/$OUT/trace -o corun -- python $R/bench.py --steps 3 --warmup 1 --no-h2h --cli-chunks 0 --no-cpu-baseline > $R/$OUT/trace_bench.json 2> $R/$OUT/trace.err
cd $R
ls -la $OUT/trace* | head
