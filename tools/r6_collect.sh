#!/bin/bash
# round 6, final build: everything refreshed under profiles/r6/ (GPU box, repo root):  bash tools/r6_collect.sh
R=$PWD; O=$R/gpurun_out/r6/final; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.txt
bash tools/collect_profiles.sh $O > $O/collect.log 2>&1
python tools/floor_table.py $O/pmc_summary.csv $O/kernel_stats_bench_steps3.csv > $O/floor_table.md 2>> $O/summary.txt
B="--steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0"
for w in "cfg3" "cfg3-heavy"; do
  n=$(echo $w | tr ' ' '_' | tr -d '-')
  timeout 600 python bench.py --workload $w $B > $O/bench_$n.log 2>&1; grep '^{' $O/bench_$n.log > $O/bench_$n.json
  python3 -c "import json; d=json.load(open('$O/bench_$n.json')); print('$w', d['config']['fragments_total'], d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/summary.txt
done
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20_driver_style.log 2>&1; grep '^{' $O/bench_steps20_driver_style.log > $O/bench_steps20_driver_style.json
python3 -c "import json; d=json.load(open('$O/bench_steps20_driver_style.json')); h=d['host_to_host']; print('driver style', d['value'], d['ms_per_step'], d['kernels_ms_per_step'], 'h2h', h['host_to_host_mbp_s'], 'text', h['as_bedgraph_gz']['host_to_host_mbp_s'], 'stale', d['roofline']['traffic_source']['stale'])" >> $O/summary.txt
timeout 900 python bench.py --workload cfg4 --chunks 60000 --steps 3 --warmup 1 --no-cpu-baseline --no-h2h --cli-chunks 0 > $O/bench_cfg4_60k.log 2>&1; grep '^{' $O/bench_cfg4_60k.log > $O/bench_cfg4_60k_tiles_one_gpu_share.json
python3 -c "import json; d=json.load(open('$O/bench_cfg4_60k_tiles_one_gpu_share.json')); print('cfg4 60k', d['value'], d['ms_per_step'])" >> $O/summary.txt
timeout 900 python bench.py --gpus 2 --share-device --steps 5 --warmup 1 --no-cpu-baseline --h2h-ranks > $O/bench_gpus2_share_device_h2h_ranks.log 2>&1; grep '^{' $O/bench_gpus2_share_device_h2h_ranks.log > $O/bench_gpus2_share_device_h2h_ranks.json
python3 -c "import json; d=json.load(open('$O/bench_gpus2_share_device_h2h_ranks.json')); print('gpus 2 shared', d['value'], [(r['rank'], r['host_to_host_mbp_s'], r['placement']['cpu_list'], r['placement']['gpu_numa_node']) for r in d['per_rank']])" >> $O/summary.txt
bash tools/r6_textz_prof.sh final > /dev/null 2>&1; cp gpurun_out/r6/textz/kernels_final.txt $O/device_writer_kernels_20k_chunks.txt
FUZZ_SECONDS=180 timeout 500 python tests/fuzz/fuzz_parity.py 100000 21 >> $O/fuzz.log 2>&1; echo "fuzz_parity rc=$?" >> $O/summary.txt
for f in fuzz_generic fuzz_round4 fuzz_writer; do FUZZ_SECONDS=90 timeout 300 python tests/fuzz/$f.py 100000 21 >> $O/fuzz.log 2>&1; echo "$f rc=$?" >> $O/summary.txt; done
tail -3 $O/pytest_gpu.log; cat $O/summary.txt; tail -4 $O/fuzz.log; cat $O/floor_table.md
