#!/bin/bash
mkdir -p gpurun_out
for lib in fcdiff_amd/libfcdiff_hip.so profiles/var_exp1.so profiles/var_exp2.so; do
for v in 0 1; do
FCDIFF_HIP_LIB=$lib FCD_R_DSPLIT=$v timeout -k 10 200 python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-vb --no-corr > gpurun_out/r03y_bench.json 2> gpurun_out/r03y_bench.err || exit 1
python3 - $lib $v <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03y_bench.json").read().strip().splitlines()[-1])
print(sys.argv[1], "dsplit knob", sys.argv[2], round(d["ms_per_step"],4), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
PY
done
done
