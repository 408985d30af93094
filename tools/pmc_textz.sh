#!/bin/bash
# On the GPU box from the repo root: kernel-trace statistics + PMC passes of the device track writer (tools/prof_textz.py)
#   bash tools/pmc_textz.sh [OUTDIR]
R=$PWD
OUT=${1:-$R/gpurun_out/prof_textz}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tzkt -o w --output-format csv -- python $R/tools/prof_textz.py > $OUT/writer.log 2>&1
cp /tmp/tzkt/w_kernel_stats.csv $OUT/kernel_stats_device_writer.csv
i=0
dirs=""
for pmc in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $pmc --kernel-trace -d /tmp/tzpmc$i -o p --output-format csv -- python $R/tools/prof_textz.py > /tmp/tzpmc$i.log 2>&1
  echo "pass $i ($pmc) rc=$?"
  dirs="$dirs /tmp/tzpmc$i"
done
python $R/tools/pmc_summarize.py $OUT/pmc_summary_writer.csv $dirs
grep "tz_\|Name" $OUT/pmc_summary_writer.csv | cut -c1-400
