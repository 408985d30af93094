#!/bin/bash
O=$PWD/gpurun_out/r6/sess5; mkdir -p $O
BINS="mb_fft_s0 mb_fft_a0 mb_fft_a0w0 mb_fft_a0k4 mb_fft_a8" bash tools/r6_skew4.sh > /dev/null 2>&1; cp gpurun_out/r6/skew4/harness.txt $O/harness.txt
grep -A1 "^== " $O/harness.txt | grep -v "^--" | paste - - | awk '{print $2, $4, $10, $11}' | tail -24
timeout 1500 python -m pytest tests/test_gpu_bg_ext.py tests/test_gpu_golden.py tests/test_gpu_properties.py tests/test_gpu_generic_params.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2h --cli-chunks 0 2>&1 | grep '^{' > $O/bench.json; python - <<'PY'
import json
d=json.load(open("gpurun_out/r6/sess5/bench.json"))
print(d["value"], d["ms_per_step"], {k:v for k,v in d.get("roofline",{}).items() if k in ("achieved","frac","kernel_ms")})
PY
