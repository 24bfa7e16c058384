# Round 2: the script that produced the r02_*_final files of profiles/ in one box session (bash profiles/r02_final_run.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/r02_gpu_tests.txt 2>&1; rc=$?; tail -2 $O/r02_gpu_tests.txt; stop_if_killed $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; rc=$?; tail -1 $O/smoke.txt; stop_if_killed $rc
for cfg in 3 5; do
  if [ $cfg = 3 ]; then A=""; else A="--nreg 400 --subjects 500 --steps 10 --warmup 2"; fi
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch$cfg -o p -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr > $O/pmc_fetch$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write$cfg -o p -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr > $O/pmc_write$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  python3 profiles/pmc_traffic.py $O/pmc_fetch$cfg $O/pmc_write$cfg profiles/r02_pmc_traffic_cfg$cfg.json > $O/pmc_traffic$cfg.log 2>&1
  cp profiles/r02_pmc_traffic_cfg$cfg.json $O/
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats$cfg -o k -- python3 bench.py $A --no-cpu-baseline --no-vb > $O/bench_cfg${cfg}_prof.json 2> $O/bench_cfg${cfg}_prof.err; rc=$?; stop_if_killed $rc
  F=$(find $O/kstats$cfg -name "*kernel_stats.csv" | head -1); cp $F $O/r02_kernel_stats_cfg${cfg}_final.csv; python3 profiles/summarize.py $F 18 > $O/r02_kernel_stats_cfg${cfg}_final.txt
  timeout -k 10 600 python3 bench.py $A > $O/r02_bench_cfg${cfg}_final.json 2> $O/bench_cfg$cfg.err; rc=$?; stop_if_killed $rc
  echo cfg$cfg done
done
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r02_bench_cfg3_500steps_final.json 2> $O/bench_500.err; rc=$?; stop_if_killed $rc
FCD_F_FORM=3 timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 5 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/r02_bench_cfg5_old_f_kernel.json 2> $O/bench_oldf.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vb > $O/pmc_lds.log 2>&1; rc=$?; stop_if_killed $rc
find $O/pmc_lds -name "*counter_collection.csv" -exec cp {} $O/pmc_lds_cfg3.csv \;
python3 profiles/pmc_lds_summary.py $O/pmc_lds_cfg3.csv "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vb   (cfg3, final build of round 2)" > $O/r02_pmc_lds_cfg3.txt
# the step-per-launch form of the r pass (knob r_path=3) beside the default (pipelined one-launch form): same bench, 500 steps
FCD_R_PATH=3 timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r02_bench_cfg3_500steps_step_form.json 2> $O/bench_500p.err; rc=$?; stop_if_killed $rc
FCD_R_PATH=3 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vb --no-corr > $O/r02_bench_cfg3_step_form.json 2>> $O/bench_500p.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 python3 profiles/fixed_cost.py > $O/r02_fixed_cost.txt 2>&1; rc=$?; stop_if_killed $rc
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/ablate_pipe.py > $O/r02_ablate_pipe.txt 2>&1; rc=$?; stop_if_killed $rc
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/r02_trace_pipe.txt 2>&1; rc=$?; stop_if_killed $rc
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/ablate.py > $O/r02_ablate.txt 2>&1; rc=$?; stop_if_killed $rc
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_r.py > $O/r02_trace_r_step.txt 2>&1; rc=$?; stop_if_killed $rc
rm -rf $O/pmc_fetch3 $O/pmc_write3 $O/pmc_fetch5 $O/pmc_write5 $O/kstats3 $O/kstats5 $O/pmc_lds
tail -c 600 $O/r02_bench_cfg3_final.json; du -sh $O
