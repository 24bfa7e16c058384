#!/bin/bash
for l in fcdiff_amd/libfcdiff_hip.so profiles/var_place1.so profiles/var_place2.so profiles/var_place3.so; do
FCDIFF_HIP_LIB=$l timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr 2>/dev/null > gpurun_out/r03ap.json || exit 1
python3 - $l <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03ap.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"],4), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d["kernels"].items()})
PY
done
