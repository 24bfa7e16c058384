# round 3, call o: full GPU test suite + bench after the pruning (ABI 3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03o; mkdir -p $O
timeout -k 10 1100 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1; echo rc=$?
tail -4 $O/gpu_tests.txt
timeout -k 10 300 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench.json 2> $O/bench.err; echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03o/bench.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()}, d["config"]["r_pass_form"])
PY
