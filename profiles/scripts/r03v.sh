#!/bin/bash
# helper waves set the rows out in an L2-resident scratch: parity, bench x2 layouts, timeline x2 layouts
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass or pipelined or random_shapes or cfg3" > gpurun_out/r03v_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03v_tests.log
for v in 0 1; do
FCD_R_DSPLIT=$v timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr > gpurun_out/r03v_bench_$v.json 2> gpurun_out/r03v_bench_$v.err || exit 1
python3 - $v <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03v_bench_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("dsplit knob", sys.argv[1], round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()}, d["config"]["r_pass_form"])
PY
FCD_R_DSPLIT=$v FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 200 python3 profiles/trace_pipe.py > gpurun_out/r03v_trace_$v.txt 2>&1 || exit 1
done
