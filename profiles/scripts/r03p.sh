#!/bin/bash
# D split: parity of the pipelined forms, then bench with and without
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass or pipelined or random_shapes or cfg3" > gpurun_out/r03p_tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03p_tests.log
timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r03p_bench_split.json 2> gpurun_out/r03p_bench_split.err && tail -1 gpurun_out/r03p_bench_split.json &&
FCD_R_DSPLIT=1 timeout -k 10 200 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/r03p_bench_one.json 2> gpurun_out/r03p_bench_one.err && tail -1 gpurun_out/r03p_bench_one.json
