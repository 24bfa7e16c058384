# round 2, call q: pipelined one-launch r pass (knob r_path=2): parity, then A/B against the step-per-launch form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02q; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "pipelined or r_pass_forms" > $O/gpu_tests_pipe.txt 2>&1; rc=$?; tail -15 $O/gpu_tests_pipe.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "pipelined tests failed: stopping"; exit 1; fi
for v in 0 2 0 2; do
  FCD_R_PATH=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_path${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02q/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()}, d["passes_ms"])
PY
tail -5 $O/bench.err
