#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "corr" > gpurun_out/r03ad_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r03ad_tests.log
timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-vb > gpurun_out/r03ad_bench.json 2> gpurun_out/r03ad_bench.err || exit 1
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03ad_bench.json").read().strip().splitlines()[-1])
print(d["corr"])
PY
