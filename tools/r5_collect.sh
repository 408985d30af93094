#!/bin/bash
# round 5: everything committed under profiles/r5/ (GPU box, repo root):  bash tools/r5_collect.sh
R=$PWD; O=$R/gpurun_out/r5/final; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" > $O/summary.txt
bash tools/collect_profiles.sh $O > $O/collect.log 2>&1
B="--steps 10 --warmup 2 --no-cpu-baseline --no-h2h --cli-chunks 0"
for w in "cfg3" "cfg3 --frags-per-chunk 545" "cfg3-heavy"; do
  n=$(echo $w | tr ' ' '_' | tr -d '-')
  timeout 600 python bench.py --workload $w $B > $O/bench_$n.log 2>&1; grep '^{' $O/bench_$n.log > $O/bench_$n.json
  python3 -c "import json; d=json.load(open('$O/bench_$n.json')); print('$w', d['config']['fragments_total'], d['value'], d['ms_per_step'], d['kernels_ms_per_step'])" >> $O/summary.txt
done
NATAC_OCC_ORDER=0 timeout 600 python bench.py --workload cfg3-heavy $B > $O/bench_cfg3heavy_chunk_order.log 2>&1; grep '^{' $O/bench_cfg3heavy_chunk_order.log > $O/bench_cfg3heavy_chunk_order.json
timeout 600 python tools/occ_params_timing.py 20000 > $O/occ_params.json 2>> $O/summary.txt
timeout 900 python bench.py --gpus 2 --share-device --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_gpus2_plain_start_share_device.log 2>&1; grep '^{' $O/bench_gpus2_plain_start_share_device.log > $O/bench_gpus2_plain_start_share_device.json
for f in fuzz_parity fuzz_generic fuzz_round4; do FUZZ_SECONDS=150 timeout 400 python tests/fuzz/$f.py 100000 5 >> $O/fuzz.log 2>&1; echo "$f rc=$?" >> $O/summary.txt; done
tail -3 $O/pytest_gpu.log; cat $O/summary.txt; tail -3 $O/fuzz.log
