# round 2, call zu: pipelined r pass, the in-order role asks for its next tiles behind the stores of the block it finishes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zu; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "pipelined or r_pass_forms or odd_shapes" > $O/gpu_tests.txt 2>&1; rc=$?; tail -2 $O/gpu_tests.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "tests failed: stopping"; exit 1; fi
for v in 0 3 0; do
  FCD_R_PATH=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_path${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zu/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/trace_pipe.txt 2>&1; tail -9 $O/trace_pipe.txt | head -8
