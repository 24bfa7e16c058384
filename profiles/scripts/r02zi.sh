# round 2, call zi: 20 timed steps after 5 / 50 / 200 / 1000 untimed ones
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zi; mkdir -p $O
for w in 5 50 200 1000 5 200; do
  timeout -k 10 300 python3 bench.py --steps 20 --warmup $w --no-cpu-baseline --no-vb --no-corr > $O/bench_w${w}_$RANDOM.json 2>> $O/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zi/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["warmup"], round(d["ms_per_step"],4))
PY
