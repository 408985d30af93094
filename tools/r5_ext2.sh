#!/bin/bash
# round 5, extended FFT tiles: counters of natac_background_edge / natac_background_fft in the harness (separate --pmc passes)
R=$PWD; O=$R/gpurun_out/r5/ext2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $pmc --kernel-trace -d /tmp/e$i -o p --output-format csv -- $R/tools/mb_fft 20000 2120 1 x > /tmp/e$i.log 2>&1
  echo "pass $i ($pmc) rc=$?"
  python3 - /tmp/e$i <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        if "background" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
    for k in acc:
        for c in acc[k]: print("  %-40s %-24s %.4g per launch (%d launches)" % (k, c, acc[k][c] / n[(k, c)], n[(k, c)]))
PY
done > $O/pmc.txt 2>&1
cat $O/pmc.txt
