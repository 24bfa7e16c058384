# Round 4: the script that produced the r04_* files of profiles/ in one box session (bash profiles/r04_final_run.sh).
# Needs both libraries and the micro-benchmarks built in tree:
#   make -C fcdiff_amd/csrc && make -C fcdiff_amd/csrc ABLATE=1 && make -C fcdiff_amd/csrc fcd_gibbs.s fcd_gibbs_r.s
#   for x in valu_rate vmem_rate; do hipcc -O3 -w --offload-arch=gfx950 profiles/micro/$x.hip -o profiles/micro/$x; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/r04_gpu_tests.txt 2>&1; rc=$?; tail -2 $O/r04_gpu_tests.txt; stop_if_killed $rc
cp gpurun_out/tie_margin_cfg3.json $O/r04_tie_margin_cfg3.json 2>/dev/null
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; rc=$?; tail -1 $O/smoke.txt; stop_if_killed $rc
CTRS="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY"
for cfg in 3 5; do
  if [ $cfg = 3 ]; then A=""; else A="--nreg 400 --subjects 500 --steps 10 --warmup 2"; fi
  # counters first (separate passes, --kernel-trace only): the bench runs below read their summaries
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch$cfg -o p -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr > $O/pmc_fetch$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write$cfg -o p -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr > $O/pmc_write$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  python3 profiles/pmc_traffic.py $O/pmc_fetch$cfg $O/pmc_write$cfg profiles/r04_pmc_traffic_cfg$cfg.json > $O/pmc_traffic$cfg.log 2>&1
  cp profiles/r04_pmc_traffic_cfg$cfg.json $O/
  timeout -k 10 400 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $O/pmc_lds$cfg -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr > $O/pmc_lds$cfg.log 2>&1; rc=$?; stop_if_killed $rc
  f=$(find $O/pmc_lds$cfg -name "*counter_collection.csv" | head -1)
  python3 profiles/pmc_lds_summary.py $f "rocprofv3 --pmc $CTRS --kernel-trace -- python3 bench.py $A --no-cpu-baseline --no-vb --no-corr   (cfg$cfg, final build of round 4)" profiles/r04_pmc_lds_cfg$cfg.json > $O/r04_pmc_lds_cfg$cfg.txt
  cp profiles/r04_pmc_lds_cfg$cfg.json $O/
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kstats$cfg -o k -- python3 bench.py $A --no-cpu-baseline --no-vb > $O/bench_cfg${cfg}_prof.json 2> $O/bench_cfg${cfg}_prof.err; rc=$?; stop_if_killed $rc
  F=$(find $O/kstats$cfg -name "*kernel_stats.csv" | head -1); cp $F $O/r04_kernel_stats_cfg${cfg}.csv; python3 profiles/summarize.py $F 18 > $O/r04_kernel_stats_cfg${cfg}.txt
  timeout -k 10 600 python3 bench.py $A > $O/r04_bench_cfg${cfg}.json 2> $O/bench_cfg$cfg.err; rc=$?; stop_if_killed $rc
  echo cfg$cfg done
done
# a long run; the several-rank loop on a process group of one rank (the library's own communicator / round 3's loop through torch)
timeout -k 10 300 python3 bench.py --steps 2000 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r04_bench_cfg3_2000steps.json 2> $O/bench_2000.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --force-pg --no-cpu-baseline --no-vb --no-corr > $O/r04_bench_cfg3_500steps_process_group_of_one.json 2> $O/bench_pg.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --force-pg --collective torch --no-cpu-baseline --no-vb --no-corr > $O/r04_bench_cfg3_500steps_process_group_of_one_torch.json 2>> $O/bench_pg.err; rc=$?; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r04_bench_cfg3_500steps.json 2>> $O/bench_pg.err; rc=$?; stop_if_killed $rc
FCD_R_COOP=1 timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r04_bench_cfg3_500steps_cooperative_launch.json 2> $O/bench_coop.err; rc=$?; stop_if_killed $rc
FCD_R_PATH=3 timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/r04_bench_cfg3_500steps_step_form.json 2> $O/bench_500p.err; rc=$?; stop_if_killed $rc
# what the RCCL kernel of the 8-word all-reduce occupies (grid, workgroup) -- on the compute stream it never runs beside the r pass
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/ktrace_pg -o t -- python3 bench.py --steps 20 --warmup 5 --force-pg --no-cpu-baseline --no-vb --no-corr > $O/ktrace_pg.json 2> $O/ktrace_pg.err; rc=$?; stop_if_killed $rc
f=$(find $O/ktrace_pg -name "*kernel_trace.csv" | head -1); python3 - "$f" > $O/r04_rccl_kernel_footprint.txt <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "nccl" in r["Kernel_Name"].lower() or "rccl" in r["Kernel_Name"].lower()]
print("rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 5 --force-pg ...: kernels of RCCL in the trace (one rank)")
c = collections.Counter((r["Kernel_Name"][:90], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))) for r in rows)
for (k, n) in c.most_common(8):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows if r["Kernel_Name"][:90] == k[0]]
    print("  %5d x  grid %s  workgroup %s  mean %.1f us   %s" % (n, k[1], k[2], sum(d) / len(d) / 1e3, k[0]))
PY
# K_corr alone, the micro-benchmarks, fixed cost per call, the cfg5 ablation
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kcorr -o corr -- python3 profiles/corr_only.py 20 > $O/r04_corr_only.txt 2>&1; rc=$?; stop_if_killed $rc
F=$(find $O/kcorr -name "*kernel_stats.csv" | head -1); python3 profiles/summarize.py $F 6 >> $O/r04_corr_only.txt
timeout -k 10 120 profiles/micro/vmem_rate > $O/r04_ubench_vmem_rate.txt 2>&1
timeout -k 10 300 python3 profiles/fixed_cost.py > $O/r04_fixed_cost.txt 2>&1; rc=$?; stop_if_killed $rc
timeout -k 10 300 python3 profiles/vb_iter.py > $O/r04_vb_iter.txt 2>&1; rc=$?; stop_if_killed $rc
rm -rf $O/pmc_fetch3 $O/pmc_write3 $O/pmc_fetch5 $O/pmc_write5 $O/kstats3 $O/kstats5 $O/pmc_lds3 $O/pmc_lds5 $O/kcorr $O/ktrace_pg
tail -c 600 $O/r04_bench_cfg3.json; du -sh $O
