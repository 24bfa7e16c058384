# round 2, call zv: the loop several ranks run (one call per M-step period, lagged M-step) on ONE rank, against the single-call loop
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zv; mkdir -p $O
for lag in 0 1 0 1; do
  timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --mstep-lag $lag --no-cpu-baseline --no-vb --no-corr > $O/bench_lag${lag}_$RANDOM.json 2>> $O/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zv/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["config"]["mstep_lag"], round(d["ms_per_step"],4), round(d["value"]))
PY
