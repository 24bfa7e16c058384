# round 2, call b: GPU tests of the new build, cfg3 / cfg5 benches, LDS counters  (bash profiles/scripts/r02b.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg3.json 2> $O/bench_cfg3.err; rc=$?; stop_if_killed $rc; echo cfg3 $rc
FCD_F_FORM=2 timeout -k 10 200 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_cfg3_fform2.json 2> $O/bench_cfg3_fform2.err; rc=$?; stop_if_killed $rc; echo cfg3-fform2 $rc
timeout -k 10 200 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline > $O/bench_cfg3_500.json 2> $O/bench_cfg3_500.err; rc=$?; stop_if_killed $rc; echo cfg3-500 $rc
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/pmc_lds -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/pmc_lds.log 2>&1; rc=$?; stop_if_killed $rc; echo pmc $rc
timeout -k 10 400 python3 bench.py --nreg 400 --subjects 500 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5.json 2> $O/bench_cfg5.err; rc=$?; stop_if_killed $rc; echo cfg5 $rc
FCD_F_FORM=3 timeout -k 10 400 python3 bench.py --nreg 400 --subjects 500 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_fform3.json 2> $O/bench_cfg5_fform3.err; rc=$?; stop_if_killed $rc; echo cfg5-old-f $rc
FCD_R_UB=1 timeout -k 10 400 python3 bench.py --nreg 400 --subjects 500 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_cfg5_ub1.json 2> $O/bench_cfg5_ub1.err; rc=$?; stop_if_killed $rc; echo cfg5-ub1 $rc
find $O/pmc_lds -name "*.csv" | head; du -sh $O
