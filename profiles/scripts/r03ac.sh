#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03ac_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03ac_tests.log
timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr > gpurun_out/r03ac_bench.json 2> gpurun_out/r03ac_bench.err || exit 1
python3 - <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03ac_bench.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()}, d["passes_ms"])
PY
