R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o bench --output-format csv -- python $R/bench.py --no-cpu-baseline --no-h2h --steps 5 > /tmp/kt.log 2>&1
python $R/tools/kstats.py /tmp/kt/bench_kernel_stats.csv | head -24
grep '^{' /tmp/kt.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
