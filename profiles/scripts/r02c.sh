# round 2, call c: GPU tests, full bench lines (cfg3, cfg5), kernel stats and PMC traffic for both
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02c; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
timeout -k 10 300 python3 bench.py > $O/bench_cfg3.json 2> $O/bench_cfg3.err; rc=$?; stop_if_killed $rc; echo cfg3 $rc
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb > $O/bench_cfg3_500.json 2> $O/bench_cfg3_500.err; rc=$?; stop_if_killed $rc; echo cfg3-500 $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats3 -o k -- python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3_prof.json 2> $O/bench_cfg3_prof.err; rc=$?; stop_if_killed $rc; echo kstats3 $rc
timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 > $O/bench_cfg5.json 2> $O/bench_cfg5.err; rc=$?; stop_if_killed $rc; echo cfg5 $rc
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats5 -o k -- python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb > $O/bench_cfg5_prof.json 2> $O/bench_cfg5_prof.err; rc=$?; stop_if_killed $rc; echo kstats5 $rc
for cfg in 3 5; do
  if [ $cfg = 3 ]; then A=""; else A="--nreg 400 --subjects 500"; fi
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch$cfg -o p -- python3 bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-vb > $O/pmc_fetch$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write$cfg -o p -- python3 bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-vb > $O/pmc_write$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  python3 profiles/pmc_traffic.py $O/pmc_fetch$cfg $O/pmc_write$cfg $O/r02_pmc_traffic_cfg$cfg.json > $O/pmc_traffic$cfg.log 2>&1
  echo pmc$cfg done
done
for d in kstats3 kstats5; do F=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $F $O/${d}_kernel_stats.csv; python3 profiles/summarize.py $F 16 > $O/${d}.txt; done
rm -rf $O/pmc_fetch3 $O/pmc_write3 $O/pmc_fetch5 $O/pmc_write5 $O/kstats3 $O/kstats5
du -sh $O
