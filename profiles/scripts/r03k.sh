cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03k; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "r_pass_forms and pipelined" > $O/tests.txt 2>&1; echo rc=$?; tail -2 $O/tests.txt
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr"
FCD_R_PATH=4 timeout -k 10 300 $B > $O/old.json 2> $O/e1; echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03k/old.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],4), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
PY
