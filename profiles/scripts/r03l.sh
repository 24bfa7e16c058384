# round 3, call l: cfg5 share (Nreg 400, U 250, 1024 chains): pipelined form with patient groups vs step-per-launch form
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l; mkdir -p $O
B="python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr"
timeout -k 10 500 $B > $O/cfg5_pipe2.json 2> $O/e0; echo rc=$?
FCD_R_PATH=3 timeout -k 10 500 $B > $O/cfg5_step.json 2> $O/e1; echo rc=$?
python3 - <<'PY'
import json
for n in ("cfg5_pipe2","cfg5_step"):
    try:
        d=json.loads(open("gpurun_out/r03l/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],4), round(d["value"]), {k:(round(v["avg_launch_ms"]*1e3,1), v["launches"]) for k,v in d.get("kernels",{}).items()})
    except Exception as e: print(n,"failed",e)
PY
tail -3 $O/e0
