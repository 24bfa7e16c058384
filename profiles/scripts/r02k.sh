# round 2, call k: K_corr with per-wave tile lists, block edge 64 vs 128
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02k; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "corr" > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
for bt in 4 8; do
FCD_CORR_BT=$bt timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vb > $O/bench_cfg3_bt$bt.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
FCD_CORR_BT=$bt timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 2 --warmup 1 --no-cpu-baseline --no-vb > $O/bench_cfg5_bt$bt.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json
for f in ("bench_cfg3_bt4","bench_cfg3_bt8","bench_cfg5_bt4","bench_cfg5_bt8"):
    d=json.loads(open("gpurun_out/r02k/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["corr"]["ms"], d["corr"]["frac"])
PY
