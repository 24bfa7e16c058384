# round 2, call zf: cfg3, second panel workgroup of every CU started 3.5 / 7 us late (knob r_stagger)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zf; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
for v in 0 1 2 0 1; do
  FCD_R_STAGGER=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_st${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zf/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
