set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r02a/bench_cfg3.json 2> gpurun_out/r02a/bench_cfg3.err
echo cfg3 done
timeout -k 10 300 python3 bench.py --steps 500 --warmup 10 --no-cpu-baseline > gpurun_out/r02a/bench_cfg3_500.json 2> gpurun_out/r02a/bench_cfg3_500.err
echo cfg3-500 done
timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/r02a/pmc_lds -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02a/pmc_lds.log 2>&1
echo pmc done
timeout -k 10 500 python3 bench.py --nreg 400 --subjects 500 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r02a/bench_cfg5.json 2> gpurun_out/r02a/bench_cfg5.err
echo cfg5 done
