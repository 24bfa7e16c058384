# round 2, call l: K_corr with MFMA accumulators kept in VGPRs (compiler flag), MFMA busy counter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "corr" > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vb > $O/bench_cfg3.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 2 --warmup 1 --no-cpu-baseline --no-vb > $O/bench_cfg5.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
true
find $O/pmc_mfma -name "*counter_collection.csv" -exec cp {} $O/pmc_mfma.csv \; ; find $O/pmc_mfma -name "*kernel_trace.csv" -exec cp {} $O/pmc_mfma_trace.csv \; ; rm -rf $O/pmc_mfma
python3 - <<'PY'
import json
for f in ("bench_cfg3","bench_cfg5"):
    d=json.loads(open("gpurun_out/r02m/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["corr"]["ms"], d["corr"]["frac"])
PY
