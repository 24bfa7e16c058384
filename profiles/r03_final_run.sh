# Round 3: the script that produced the r03_* files of profiles/ in one box session (bash profiles/r03_final_run.sh).
# Needs both libraries built in tree:  make -C fcdiff_amd/csrc && make -C fcdiff_amd/csrc ABLATE=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/r03_gpu_tests.txt 2>&1; rc=$?; tail -2 $O/r03_gpu_tests.txt; stop_if_killed $rc
cp gpurun_out/tie_margin_cfg3.json $O/r03_tie_margin_cfg3.json 2>/dev/null
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; rc=$?; tail -1 $O/smoke.txt; stop_if_killed $rc
for cfg in 3 5; do
  if [ $cfg = 3 ]; then A=""; else A="--nreg 400 --subjects 500 --steps 10 --warmup 2"; fi
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch$cfg -o p -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr > $O/pmc_fetch$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write$cfg -o p -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr > $O/pmc_write$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  python3 profiles/pmc_traffic.py $O/pmc_fetch$cfg $O/pmc_write$cfg profiles/r03_pmc_traffic_cfg$cfg.json > $O/pmc_traffic$cfg.log 2>&1
  cp profiles/r03_pmc_traffic_cfg$cfg.json $O/
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats$cfg -o k -- python3 bench.py $A --no-cpu-baseline --no-vb > $O/bench_cfg${cfg}_prof.json 2> $O/bench_cfg${cfg}_prof.err; rc=$?; stop_if_killed $rc
  F=$(find $O/kstats$cfg -name "*kernel_stats.csv" | head -1); cp $F $O/r03_kernel_stats_cfg${cfg}.csv; python3 profiles/summarize.py $F 18 > $O/r03_kernel_stats_cfg${cfg}.txt
  timeout -k 10 600 python3 bench.py $A > $O/r03_bench_cfg${cfg}.json 2> $O/bench_cfg$cfg.err; rc=$?; stop_if_killed $rc
  echo cfg$cfg done
done
# a long run, the several-rank loop on a process group of one rank, the other forms beside the default
timeout -k 10 300 python3 bench.py --steps 2000 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r03_bench_cfg3_2000steps.json 2> $O/bench_2000.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --force-pg --no-cpu-baseline --no-vb --no-corr > $O/r03_bench_cfg3_500steps_process_group_of_one.json 2> $O/bench_pg.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --force-pg --mstep-lag 1 --no-cpu-baseline --no-vb --no-corr > $O/r03_bench_cfg3_500steps_process_group_of_one_lagged.json 2>> $O/bench_pg.err; rc=$?; stop_if_killed $rc
FCD_R_DSPLIT=1 timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r03_bench_cfg3_500steps_one_inorder_workgroup.json 2> $O/bench_ds.err; rc=$?; stop_if_killed $rc
FCD_R_PATH=3 timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r03_bench_cfg3_500steps_step_form.json 2> $O/bench_500p.err; rc=$?; stop_if_killed $rc
FCD_CORR_FORM=1 timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-vb > $O/r03_bench_cfg3_corr_block_kernel.json 2> $O/bench_corr1.err; rc=$?; stop_if_killed $rc
# counters: LDS / VALU of the sweep kernels, the MFMA pipe of K_corr
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_lds -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vb > $O/pmc_lds.log 2>&1; rc=$?; stop_if_killed $rc
find $O/pmc_lds -name "*counter_collection.csv" -exec cp {} $O/pmc_lds_cfg3.csv \;
python3 profiles/pmc_lds_summary.py $O/pmc_lds_cfg3.csv "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY --kernel-trace -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vb   (cfg3, final build of round 3)" > $O/r03_pmc_lds_cfg3.txt
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $O/pmc_corr -o corr -- python3 profiles/corr_only.py 5 > $O/pmc_corr.log 2>&1; rc=$?; stop_if_killed $rc
f=$(find $O/pmc_corr -name "*counter_collection.csv" | head -1); python3 - "$f" > $O/r03_pmc_corr.txt <<'PY'
import csv, sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:70]; d[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[(k, r["Counter_Name"])] += 1
print("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT -- python3 profiles/corr_only.py 5   (sums over the 6 launches, S=100 Nreg=200 T=1200)")
for k, v in d.items():
    if "corr" in k: print(k); [print("    %-28s %.4g" % (c, x)) for c, x in sorted(v.items())]
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kcorr -o corr -- python3 profiles/corr_only.py 20 > $O/r03_corr_only.txt 2>&1; rc=$?; stop_if_killed $rc
F=$(find $O/kcorr -name "*kernel_stats.csv" | head -1); python3 profiles/summarize.py $F 6 >> $O/r03_corr_only.txt
timeout -k 10 120 profiles/micro/valu_rate > $O/r03_ubench_valu_rate.txt 2>&1
timeout -k 10 120 profiles/micro/mfma_f64_rate > $O/r03_ubench_mfma_f64_rate.txt 2>&1
timeout -k 10 300 python3 profiles/fixed_cost.py > $O/r03_fixed_cost.txt 2>&1; rc=$?; stop_if_killed $rc
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/ablate_pipe.py > $O/r03_ablate_pipe.txt 2>&1; rc=$?; stop_if_killed $rc
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/r03_trace_pipe.txt 2>&1; rc=$?; stop_if_killed $rc
FCD_R_DSPLIT=1 FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/r03_trace_pipe_one_inorder_workgroup.txt 2>&1; rc=$?; stop_if_killed $rc
rm -rf $O/pmc_fetch3 $O/pmc_write3 $O/pmc_fetch5 $O/pmc_write5 $O/kstats3 $O/kstats5 $O/pmc_lds $O/pmc_corr $O/kcorr
tail -c 700 $O/r03_bench_cfg3.json; du -sh $O
