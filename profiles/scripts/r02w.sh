# round 2, call w: pipelined r pass, prefetch depth of the in-order scan (builds with PF_E/PF_F = 3/2, 4/2, 4/3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02w; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "pipelined" > $O/gpu_tests_pipe.txt 2>&1; rc=$?; tail -3 $O/gpu_tests_pipe.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "pipelined tests failed: stopping"; exit 1; fi
for lib in fcdiff_amd/libfcdiff_hip.so fcdiff_amd/libfcdiff_hip_pf32.so fcdiff_amd/libfcdiff_hip_pf42.so fcdiff_amd/libfcdiff_hip.so; do
  FCDIFF_HIP_LIB=$lib FCD_R_PATH=2 timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_$(basename $lib .so)_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
FCD_R_PATH=0 timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_stepform.json 2>> $O/bench.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02w/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
