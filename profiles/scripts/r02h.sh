# round 2, call h: pair-record table for the r pass: tests, A/B at cfg3 (300 steps) and cfg5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02h; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
for v in 0 1 0 1; do
  FCD_R_NOPRE=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_nopre${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
for v in 0 1; do
  FCD_R_NOPRE=$v timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_nopre$v.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
FCD_R_UB=2 timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_pre_ub2.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats3 -o k -- python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3_prof.json 2> $O/bench_cfg3_prof.err; rc=$?; stop_if_killed $rc
F=$(find $O/kstats3 -name "*kernel_stats.csv" | head -1); python3 profiles/summarize.py $F 18 > $O/kstats3.txt; rm -rf $O/kstats3
du -sh $O
