#!/bin/bash
# round 6: the whole GPU suite on the current build with the achieved differences of every assert_track logged (tests/helpers.py, NATAC_TRACK_STATS)
O=$PWD/gpurun_out/r6/tight; mkdir -p $O; rm -f $O/stats.tsv
NATAC_TRACK_STATS=$O/stats.tsv timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $O/pytest_gpu.log
cat $O/pytest_gpu.log
sort -u $O/stats.tsv | awk -F'\t' '{print $1}' | sort | uniq -c | sort -rn | head -50
