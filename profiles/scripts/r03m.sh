# round 3, call m: reads in batches of eight + tree of additions in the panel role of the step form and of the round-2 pipelined form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03m; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass_forms" > $O/tests.txt 2>&1; echo rc=$?; tail -2 $O/tests.txt
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr"
FCD_R_PATH=4 timeout -k 10 300 $B > $O/old.json 2> $O/e1; echo rc=$?
FCD_R_PATH=3 timeout -k 10 300 $B > $O/step.json 2> $O/e2; echo rc=$?
FCD_R_PATH=3 timeout -k 10 500 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/cfg5_step.json 2> $O/e3; echo rc=$?
python3 - <<'PY'
import json
for n in ("old","step","cfg5_step"):
    d=json.loads(open("gpurun_out/r03m/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],4), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
PY
