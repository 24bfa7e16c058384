#!/bin/bash
mkdir -p gpurun_out
FCD_R_LOCAL=1 timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass or pipelined or random_shapes or cfg3 or sentinels" > gpurun_out/r03ao_tests.log 2>&1
echo "tests (r_local) rc=$?"; tail -2 gpurun_out/r03ao_tests.log
for v in 0 1; do
FCD_R_LOCAL=$v timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr > gpurun_out/r03ao_bench.json 2> gpurun_out/r03ao_bench.err || exit 1
python3 - $v <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03ao_bench.json").read().strip().splitlines()[-1])
print("r_local", sys.argv[1], round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
PY
done
