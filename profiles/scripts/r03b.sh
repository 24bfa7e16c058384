# round 3, call b: first run of the pipelined form with tagged hand-overs (pipe2): parity tests that touch the r pass + bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass_forms or pipelined or random_shapes or odd_shapes or state_for_state" > $O/tests.txt 2>&1; echo rc=$?
tail -5 $O/tests.txt
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr"
timeout -k 10 300 $B > $O/pipe2.json 2> $O/pipe2.err; echo rc=$?
FCD_R_PATH=4 timeout -k 10 300 $B > $O/pipe_old.json 2> $O/e1; echo rc=$?
python3 - <<'PY'
import json
for n in ("pipe2","pipe_old"):
    try:
        d=json.loads(open("gpurun_out/r03b/%s.json"%n).read().strip().splitlines()[-1])
        print(n, round(d["ms_per_step"],4), round(d["value"]), json.dumps(d.get("kernels",{}))[:500])
    except Exception as e: print(n, "failed", e)
PY
