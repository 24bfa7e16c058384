#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass or pipelined or random_shapes or cfg3" > gpurun_out/r03w_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03w_tests.log
for lib in fcdiff_amd/libfcdiff_hip.so profiles/var_pfs5.so; do
for v in 0; do
FCDIFF_HIP_LIB=$lib FCD_R_DSPLIT=$v timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr > gpurun_out/r03w_bench.json 2> gpurun_out/r03w_bench.err || exit 1
python3 - $lib $v <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03w_bench.json").read().strip().splitlines()[-1])
print(sys.argv[1], "dsplit knob", sys.argv[2], round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()}, d["config"]["r_pass_form"])
PY
done
done
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 200 python3 profiles/trace_pipe.py > gpurun_out/r03w_trace_0.txt 2>&1
