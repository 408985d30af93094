#!/bin/bash
# round 6: skewed loop with / without the ping-pong of the carried factor; shader clock of each variant (GRBM_GUI_ACTIVE / 8 / duration)
O=$PWD/gpurun_out/r6/sess6; mkdir -p $O
T=$PWD/tools
BINS="mb_fft_s0 mb_fft_a0pp0 mb_fft_a0 mb_fft_a0k4" bash tools/r6_skew4.sh > /dev/null 2>&1; cp gpurun_out/r6/skew4/harness.txt $O/harness.txt
grep -E "fnv|lower" $O/harness.txt
grep -A1 "^== " $O/harness.txt | grep -v "^--" | paste - - | awk '{print $2, $4, $10, $11}' | tail -16
cd /tmp && export TMPDIR=/tmp
for b in mb_fft_s0 mb_fft_a0 mb_fft_a3 mb_fft_a15; do
  rm -rf /tmp/pmc_$b
  timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_$b -o p --output-format csv -- $T/$b 20000 2120 1 x > /tmp/pmc_$b.log 2>&1
  python3 - $b <<'PY' >> $O/clocks.txt
import csv,glob,sys
b=sys.argv[1]
for f in glob.glob("/tmp/pmc_%s/**/*counter_collection.csv"%b,recursive=True):
    for r in csv.DictReader(open(f)):
        if "natac_background_fft" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE":
            ns=float(r["End_Timestamp"])-float(r["Start_Timestamp"]); print(b,"ms %.3f clock_ghz %.3f"%(ns/1e6,float(r["Counter_Value"])/8/ns))
PY
done
cat $O/clocks.txt
