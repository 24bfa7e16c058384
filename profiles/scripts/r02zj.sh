# round 2, call zj: f pass with triple records (knob f_form=4): parity, then A/B at cfg3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zj; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "triple or r_pass_forms" > $O/gpu_tests_tri.txt 2>&1; rc=$?; tail -12 $O/gpu_tests_tri.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "triple tests failed: stopping"; exit 1; fi
for v in 0 4 0 4; do
  FCD_F_FORM=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_form${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zj/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
tail -3 $O/bench.err
