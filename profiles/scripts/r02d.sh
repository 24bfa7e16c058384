# round 2, call d: tests of the K_lik / K_corr / pairx / tally changes, microbenches, bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt; stop_if_killed $rc
timeout -k 10 120 ./profiles/micro/mfma64 > $O/ubench_mfma64.txt 2>&1; rc=$?; stop_if_killed $rc; cat $O/ubench_mfma64.txt
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3.json 2> $O/bench_cfg3.err; rc=$?; stop_if_killed $rc; echo cfg3 $rc
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats3 -o k -- python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3_prof.json 2> $O/bench_cfg3_prof.err; rc=$?; stop_if_killed $rc; echo kstats3 $rc
timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb > $O/bench_cfg5.json 2> $O/bench_cfg5.err; rc=$?; stop_if_killed $rc; echo cfg5 $rc
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats5 -o k -- python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb > $O/bench_cfg5_prof.json 2> $O/bench_cfg5_prof.err; rc=$?; stop_if_killed $rc; echo kstats5 $rc
A="--nreg 400 --subjects 500"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch5 -o p -- python3 bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/pmc_fetch5.log 2>&1; rc=$?; stop_if_killed $rc
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write5 -o p -- python3 bench.py $A --steps 5 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/pmc_write5.log 2>&1; rc=$?; stop_if_killed $rc
python3 profiles/pmc_traffic.py $O/pmc_fetch5 $O/pmc_write5 $O/r02_pmc_traffic_cfg5.json > $O/pmc_traffic5.log 2>&1
for d in kstats3 kstats5; do F=$(find $O/$d -name "*kernel_stats.csv" | head -1); cp $F $O/${d}_kernel_stats.csv; python3 profiles/summarize.py $F 18 > $O/${d}.txt; done
rm -rf $O/pmc_fetch5 $O/pmc_write5 $O/kstats3 $O/kstats5
du -sh $O
