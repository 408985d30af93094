#!/bin/bash
# Run on the GPU box from the repo root:  bash tools/collect_profiles.sh [OUTDIR]
# Kernel-trace statistics of the default bench, separate PMC passes (each under its own timeout), the default bench line.
R=$PWD
OUT=${1:-$R/gpurun_out/prof}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --no-h2h --cli-chunks 0 > /tmp/kt.log 2>&1
cp /tmp/kt/bench_kernel_stats.csv $OUT/kernel_stats_bench_steps3.csv
grep '^{' /tmp/kt.log > $OUT/bench_under_rocprof.json
i=0
dirs=""
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $pmc --kernel-trace -d /tmp/pmc$i -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-h2h --cli-chunks 0 > /tmp/pmc$i.log 2>&1
  echo "pass $i ($pmc) rc=$?"
  dirs="$dirs /tmp/pmc$i"
done
python $R/tools/pmc_summarize.py $OUT/pmc_summary.csv $dirs
cd $R && timeout 400 python bench.py > $OUT/bench_default.log 2>&1; grep '^{' $OUT/bench_default.log > $OUT/bench_default.json
tail -c 400 $OUT/bench_default.json
