# round 2, call n: non-temporal stores of the square f copy (does packing get faster?), K_corr transpose epilogue parity
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02n; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
for v in 0 1 0 1; do
  FCD_F_NT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks_$v -o k -- python3 bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-vb --no-corr > $O/bench_nt${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
  F=$(find $O/ks_$v -name "*kernel_stats.csv" | head -1); python3 profiles/summarize.py $F 6 > $O/ks_nt${v}_$RANDOM.txt; rm -rf $O/ks_$v
done
cat $O/ks_nt*.txt; python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02n/bench_nt*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"])
PY
