# round 2, call g: tests, cfg3 bench with / without the r prefetch hint (300 steps each for a steady figure)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02g; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
for v in 0 1 0 1; do
  FCD_R_PREFETCH=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_pf${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
for v in 0 1; do
  FCD_R_PREFETCH=$v timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_pf$v.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb > $O/bench_cfg5.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
du -sh $O
