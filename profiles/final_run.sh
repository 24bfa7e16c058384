set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python3 -m pytest tests -m gpu -q > gpurun_out/final/r01_gpu_tests.txt 2>&1
tail -2 gpurun_out/final/r01_gpu_tests.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_fetch -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/final/pmc_write -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/final/pmc_write.log 2>&1
python3 profiles/pmc_traffic.py gpurun_out/final/pmc_fetch gpurun_out/final/pmc_write profiles/r01_pmc_traffic.json > gpurun_out/final/pmc_traffic.log 2>&1
cp profiles/r01_pmc_traffic.json gpurun_out/final/
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/kstats -o k -- python3 bench.py --no-cpu-baseline > gpurun_out/final/bench_prof.json 2> gpurun_out/final/bench_prof.err
F=$(find gpurun_out/final/kstats -name "*kernel_stats.csv" | head -1); cp $F gpurun_out/final/r01_kernel_stats_final.csv; python3 profiles/summarize.py $F 16 > gpurun_out/final/r01_kernel_stats_final.txt
python3 bench.py > gpurun_out/final/r01_bench_final.json 2> gpurun_out/final/bench.err
tail -1 gpurun_out/final/r01_bench_final.json | cut -c1-600
rm -rf gpurun_out/final/pmc_fetch/*/ gpurun_out/final/kstats/*_trace.csv 2>/dev/null
du -sh gpurun_out/final
