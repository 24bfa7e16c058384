# round 2, call o: panel records straight from the table rows (knob r_direct): tests, A/B at cfg3 / cfg5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02o; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
for v in 0 1 0 1; do
  FCD_R_DIRECT=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_d${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
for v in 0 1; do
  FCD_R_DIRECT=$v timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_d$v.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02o/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["ms_per_step"],4), round(d["kernels"]["gibbs_r_step_kernel"]["avg_launch_ms"]*1e3,2))
PY
