# round 3, call f: the roles of the pipelined pass alone, PRODUCT build (knob r_dbg; results of those runs are wrong by design)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
B="python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-vb --no-corr"
for d in 0 1 3 11 27 5 37 6; do
  FCD_R_DBG=$d timeout -k 10 300 $B > $O/dbg$d.json 2> $O/err$d; echo -n "dbg=$d rc=$? "
  python3 - $d <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r03f/dbg%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print(round(d["ms_per_step"],4), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
PY
done
