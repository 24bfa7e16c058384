# round 2, call l: K_corr with MFMA accumulators kept in VGPRs (compiler flag), MFMA busy counter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02l; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "corr" > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vb > $O/bench_cfg3.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 2 --warmup 1 --no-cpu-baseline --no-vb > $O/bench_cfg5.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 bench.py --nreg 400 --subjects 500 --steps 2 --warmup 1 --no-cpu-baseline --no-vb > $O/pmc_mfma.log 2>&1; rc=$?; stop_if_killed $rc
find $O/pmc_mfma -name "*counter_collection.csv" -exec cp {} $O/pmc_mfma.csv \; ; find $O/pmc_mfma -name "*kernel_trace.csv" -exec cp {} $O/pmc_mfma_trace.csv \; ; rm -rf $O/pmc_mfma
python3 - <<'PY'
import json
for f in ("bench_cfg3","bench_cfg5"):
    d=json.loads(open("gpurun_out/r02l/%s.json"%f).read().strip().splitlines()[-1])
    print(f, d["corr"]["ms"], d["corr"]["frac"])
PY
