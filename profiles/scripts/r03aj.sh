#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass or pipelined or random_shapes or cfg3" > gpurun_out/r03aj_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/r03aj_tests.log
timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr > gpurun_out/r03aj_bench.json 2> gpurun_out/r03aj_bench.err || exit 1
python3 - <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03aj_bench.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
PY
