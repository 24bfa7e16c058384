# round 2, call f: micro-benchmark with clock / concurrency check, tests, benches after the K_lik / K_corr grid changes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02f; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 120 ./profiles/micro/ubench > $O/ubench.txt 2>&1; rc=$?; stop_if_killed $rc; cat $O/ubench.txt
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-vb > $O/bench_cfg3.json 2> $O/bench_cfg3.err; rc=$?; stop_if_killed $rc; echo cfg3 $rc
timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb > $O/bench_cfg5.json 2> $O/bench_cfg5.err; rc=$?; stop_if_killed $rc; echo cfg5 $rc
du -sh $O
