#!/bin/bash
# rocprofv3 evidence for the device BAM decoder (bamdev_inflate / bamdev_walk): kernel-trace statistics and separate PMC passes on a
# synthetic 20 M-record BAM, then tools/bench_bam.py on 60 M records.   usage (GPU box): bash tools/prof_bam.sh [OUTDIR]
R=$PWD
OUT=${1:-$R/gpurun_out/bam}
case $OUT in /*) ;; *) OUT=$R/$OUT ;; esac
mkdir -p $OUT
BAM=/dev/shm/natac_prof.bam
python tools/bam_decode_once.py $BAM make ${N_PROF:-20000000} > $OUT/make.log 2>&1
python tools/bam_decode_once.py $BAM 3 > $OUT/decode_plain.log 2>&1; cat $OUT/decode_plain.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/ktb -o bam --output-format csv -- python $R/tools/bam_decode_once.py $BAM 2 > /tmp/ktb.log 2>&1
cp /tmp/ktb/bam_kernel_stats.csv $OUT/kernel_stats_bam_device_20M_records.csv
i=0
dirs=""
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_SALU" "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $pmc --kernel-trace -d /tmp/pmcb$i -o p --output-format csv -- python $R/tools/bam_decode_once.py $BAM 1 > /tmp/pmcb$i.log 2>&1
  echo "pass $i ($pmc) rc=$?"
  dirs="$dirs /tmp/pmcb$i"
done
python $R/tools/pmc_summarize.py $OUT/pmc_summary_bam_device.csv $dirs
rm -f $BAM
cd $R
timeout 900 python tools/bench_bam.py ${N_BENCH:-60000000} > $OUT/bench_bam_60M_records.log 2>&1; tail -8 $OUT/bench_bam_60M_records.log
