# round 2, call zr: last check of the committed state -- full GPU suite, smoke, the driver's bench command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zr; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q > $O/r02_gpu_tests.txt 2>&1; rc=$?; tail -2 $O/r02_gpu_tests.txt; stop_if_killed $rc
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; rc=$?; tail -1 $O/smoke.txt; stop_if_killed $rc
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench.err; rc=$?; stop_if_killed $rc
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r02zr/bench_driver_cmd.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["vb_iteration"]["gpu_ms"], d["allocs_in_timed_region"])
PY
