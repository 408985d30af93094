#!/bin/bash
# round 6: pair loop as it was (s0) / skewed, template spectrum requested after S1 (s1k3) / after S1's products (s1k4): timing, then counters
O=$PWD/gpurun_out/r6/skew2; mkdir -p $O
T=$PWD/tools
{
for b in $BINS; do echo "== $b variant=1 (accuracy vs direct + output hash)"; timeout 300 $T/$b 20000 2120 1 | grep -E "^FFT|bg: max rel|fnv"; done
for rep in 1 2 3 4 5; do for b in $BINS; do echo "== $b variant=1"; timeout 120 $T/$b 20000 2120 1 x | grep -E "^FFT"; done; done
} > $O/harness.txt 2>&1
cat $O/harness.txt
cd /tmp && export TMPDIR=/tmp
for b in $PMCBINS; do
  i=0
  for pmc in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVES SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rm -rf /tmp/pmc_$b$i
    timeout 200 rocprofv3 --pmc $pmc --kernel-trace -d /tmp/pmc_$b$i -o p --output-format csv -- $T/$b 20000 2120 1 x > /tmp/pmc_$b$i.log 2>&1
    echo "pass $b $i rc=$?"
  done
  python3 - $b <<'PY' >> $O/counters.txt
import csv,glob,sys,collections
b=sys.argv[1]; last=collections.OrderedDict()
for i in range(1,5):
    for f in glob.glob("/tmp/pmc_%s%d/**/*counter_collection.csv"%(b,i),recursive=True):
        for r in csv.DictReader(open(f)):
            if "natac_background_fft" not in r["Kernel_Name"]: continue
            last[r["Counter_Name"]]=float(r["Counter_Value"])
            if r["Counter_Name"]=="GRBM_GUI_ACTIVE": last["kernel_ns"]=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
print(b," ".join("%s=%.4g"%(k,v) for k,v in last.items()))
if "GRBM_GUI_ACTIVE" in last: print(b,"clock_ghz=%.3f"%(last["GRBM_GUI_ACTIVE"]/8/last["kernel_ns"]))
PY
done
cat $O/counters.txt
