# round 2, call zk: pipelined r pass as the default -- full GPU suite, smoke, cfg3 bench default / step form, cfg5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zk; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "tests failed: stopping"; exit 1; fi
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; rc=$?; tail -1 $O/smoke.txt; stop_if_killed $rc
for v in 0 3 0 3; do
  FCD_R_PATH=$v timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_path${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5.json 2>> $O/bench.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zk/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), round(d["value"]), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()}, d["roofline"]["kernel"], round(d["roofline"]["frac"],4), d["lds_roofline"]["frac"])
PY
