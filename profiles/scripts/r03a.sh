# round 3, call a: baseline of the round-2 build on this box + step form with 1 / 2 / 4 patients per panel workgroup
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr"
timeout -k 10 300 $B > $O/base.json 2> $O/base.err; echo rc=$?
FCD_R_PATH=3 timeout -k 10 300 $B > $O/step_ub2.json 2> $O/e1; echo rc=$?
FCD_R_PATH=3 FCD_R_UB=4 timeout -k 10 300 $B > $O/step_ub4.json 2> $O/e2; echo rc=$?
FCD_R_PATH=3 FCD_R_UB=1 timeout -k 10 300 $B > $O/step_ub1.json 2> $O/e3; echo rc=$?
python3 - <<'PY'
import json
for n in ("base","step_ub2","step_ub4","step_ub1"):
    d=json.loads(open("gpurun_out/r03a/%s.json"%n).read().strip().splitlines()[-1])
    print(n, round(d["ms_per_step"],4), round(d["value"]), json.dumps(d.get("kernels",{}))[:600])
PY
