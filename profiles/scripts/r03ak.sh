#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03ak_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03ak_tests.log
for a in "--force-pg" "--force-pg --mstep-lag 1" ""; do
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 $a --no-cpu-baseline --no-vb --no-corr > gpurun_out/r03ak_bench.json 2> gpurun_out/r03ak_bench.err || exit 1
python3 - "$a" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03ak_bench.json").read().strip().splitlines()[-1])
print(repr(sys.argv[1]), round(d["ms_per_step"],4), round(d["value"]), d["config"].get("process_group"), d["config"].get("allreduce_us"))
PY
done
