# round 2, call zw: 5000 timed sweeps with the pipelined r pass (a device-side wait that was given up would fail the next call)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zw; mkdir -p $O
timeout -k 10 600 python3 bench.py --steps 5000 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_5000.json 2> $O/bench.err; echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r02zw/bench_5000.json").read().strip().splitlines()[-1])
print(d["steps"], round(d["ms_per_step"],4), round(d["value"]), d["roofline"]["kernel"])
PY
tail -2 $O/bench.err
