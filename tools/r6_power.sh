#!/bin/bash
# round 6: is natac_background_fft power-limited?  hwmon power / shader clock sampled at ~50 Hz under ~6 s of back-to-back launches
O=gpurun_out/r6/power; mkdir -p $O
H=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
echo "hwmon: $H" > $O/power.txt
ls $H >> $O/power.txt 2>&1
for f in power1_cap power1_cap_max power1_cap_default power1_average power1_input freq1_input freq2_input temp1_input; do [ -r $H/$f ] && echo "$f $(cat $H/$f)" >> $O/power.txt; done
(rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | head -60) >> $O/power.txt
sample() {  # $1 = tag
  for i in $(seq 1 400); do
    p=$(cat $H/power1_average 2>/dev/null || cat $H/power1_input 2>/dev/null); f=$(cat $H/freq1_input 2>/dev/null); t=$(cat $H/temp1_input 2>/dev/null)
    echo "$1 $i $p $f $t"; sleep 0.02
  done
}
for b in ${BINS:-mb_fft_s0 mb_fft_s1}; do
  NATAC_HARNESS_REPS=${REPS:-800} timeout 300 tools/$b 20000 2120 1 x > $O/run_$b.txt 2>&1 &
  pid=$!
  sleep 2.5    # allocation + upload + direct-kernel skip; the launches start after ~2 s
  sample $b >> $O/samples.txt
  wait $pid
  grep -E "^FFT" $O/run_$b.txt >> $O/power.txt
done
python3 - <<'PY' >> gpurun_out/r6/power/power.txt
import collections
d=collections.defaultdict(list)
for l in open("gpurun_out/r6/power/samples.txt"):
    t=l.split()
    if len(t)>=4 and t[2].isdigit() and t[3].isdigit(): d[t[0]].append((int(t[2])/1e6,int(t[3])/1e6))
for k,v in d.items():
    v2=sorted(v,key=lambda x:-x[0])[:len(v)//2]   # the busy half
    print(k,"samples",len(v),"busy-half mean power W %.0f  clock MHz %.0f  max power %.0f  min clock in busy half %.0f"%(sum(a for a,b in v2)/len(v2),sum(b for a,b in v2)/len(v2),max(a for a,b in v),min(b for a,b in v2)))
PY
cat $O/power.txt
