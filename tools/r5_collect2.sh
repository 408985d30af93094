#!/bin/bash
# round 5, after the extended background tiles: everything refreshed under profiles/r5/ (GPU box, repo root):  bash tools/r5_collect2.sh
R=$PWD; O=$R/gpurun_out/r5/final2; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.txt
bash tools/collect_profiles.sh $O > $O/collect.log 2>&1
python tools/floor_table.py $O/pmc_summary.csv $O/kernel_stats_bench_steps3.csv > $O/floor_table.md 2>> $O/summary.txt
B="--steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0"
for w in "cfg3" "cfg3 --frags-per-chunk 545" "cfg3-heavy"; do
  n=$(echo $w | tr ' ' '_' | tr -d '-')
  timeout 600 python bench.py --workload $w $B > $O/bench_$n.log 2>&1; grep '^{' $O/bench_$n.log > $O/bench_$n.json
  python3 -c "import json; d=json.load(open('$O/bench_$n.json')); print('$w', d['config']['fragments_total'], d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/summary.txt
done
NATAC_BG_EXT=0 timeout 600 python bench.py --workload cfg3 $B > $O/bench_cfg3_plain_tiles.log 2>&1; grep '^{' $O/bench_cfg3_plain_tiles.log > $O/bench_cfg3_plain_tiles.json
python3 -c "import json; d=json.load(open('$O/bench_cfg3_plain_tiles.json')); print('cfg3 NATAC_BG_EXT=0', d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/summary.txt
timeout 900 python bench.py --workload cfg4 --chunks 60000 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --cli-chunks 0 > $O/bench_cfg4_60k.log 2>&1; grep '^{' $O/bench_cfg4_60k.log > $O/bench_cfg4_60k_tiles_one_gpu_share.json
python3 -c "import json; d=json.load(open('$O/bench_cfg4_60k_tiles_one_gpu_share.json')); print('cfg4 60k', d['value'], d['ms_per_step'])" >> $O/summary.txt
timeout 900 python bench.py --gpus 2 --share-device --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_gpus2_plain_start_share_device.log 2>&1; grep '^{' $O/bench_gpus2_plain_start_share_device.log > $O/bench_gpus2_plain_start_share_device.json
FUZZ_SECONDS=240 timeout 500 python tests/fuzz/fuzz_parity.py 100000 21 >> $O/fuzz.log 2>&1; echo "fuzz_parity rc=$?" >> $O/summary.txt
for f in fuzz_generic fuzz_round4; do FUZZ_SECONDS=90 timeout 300 python tests/fuzz/$f.py 100000 21 >> $O/fuzz.log 2>&1; echo "$f rc=$?" >> $O/summary.txt; done
tail -3 $O/pytest_gpu.log; cat $O/summary.txt; tail -3 $O/fuzz.log; cat $O/floor_table.md
