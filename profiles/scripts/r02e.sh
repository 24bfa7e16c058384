# round 2, call e: tests (two-stream r pass, K_corr loads), cfg3/cfg5 benches with and without r_streams=2, K_lik counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02e; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -5 $O/gpu_tests.txt; stop_if_killed $rc
for v in 1 2; do
  FCD_R_STREAMS=$v timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3_rs$v.json 2> $O/bench_cfg3_rs$v.err; rc=$?; stop_if_killed $rc; echo cfg3 rs$v $rc
  FCD_R_STREAMS=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_300_rs$v.json 2> $O/bench_cfg3_300_rs$v.err; rc=$?; stop_if_killed $rc
  FCD_R_STREAMS=$v timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb > $O/bench_cfg5_rs$v.json 2> $O/bench_cfg5_rs$v.err; rc=$?; stop_if_killed $rc; echo cfg5 rs$v $rc
done
FCD_R_STREAMS=2 FCD_R_NOPAD=1 timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_300_rs2b.json 2> $O/bench_cfg3_300_rs2b.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/pmc_valu -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vb > $O/pmc_valu.log 2>&1; rc=$?; stop_if_killed $rc; echo pmc $rc
FCD_R_STREAMS=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats3 -o k -- python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3_prof.json 2> $O/bench_cfg3_prof.err; rc=$?; stop_if_killed $rc
F=$(find $O/kstats3 -name "*kernel_stats.csv" | head -1); python3 profiles/summarize.py $F 18 > $O/kstats3_rs2.txt; rm -rf $O/kstats3
find $O/pmc_valu -name "*counter_collection.csv" -exec cp {} $O/pmc_valu.csv \; ; rm -rf $O/pmc_valu
du -sh $O
