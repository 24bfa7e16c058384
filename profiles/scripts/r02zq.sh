# round 2, call zq: pipelined r pass, contiguous pieces of the (chunk, row) list per XCD (knob r_xcd): parity, A/B, traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zq; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "r_pass_forms" > $O/gpu_tests.txt 2>&1; rc=$?; tail -2 $O/gpu_tests.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "tests failed: stopping"; exit 1; fi
for v in 0 1 0 1; do
  FCD_R_XCD=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_xcd${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zq/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
FCD_R_XCD=1 timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --no-cpu-baseline --no-vb --no-corr > $O/pmc_fetch.log 2>&1; rc=$?; stop_if_killed $rc
FCD_R_XCD=1 timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- python3 bench.py --no-cpu-baseline --no-vb --no-corr > $O/pmc_write.log 2>&1; rc=$?; stop_if_killed $rc
python3 profiles/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_xcd1.json | grep "r_pipe\|f_pair"
rm -rf $O/pmc_fetch $O/pmc_write
