# round 2, call p: staging loops with two loads in flight (P, D, f pair), D first loads before staging, pack_f two blocks
# per wave, prefetch hints that do not wait (r_prefetch bit 0: next step's table rows; bit 1: the D scan's lines)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02p; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "tests failed: stopping"; exit 1; fi
for v in 0 3 1 2 0 3; do
  FCD_R_PREFETCH=$v timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_pf${v}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
for v in 0 3; do
  FCD_R_PREFETCH=$v timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_pf$v.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
for v in 0 3; do
  FCD_R_PREFETCH=$v FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_r.py > $O/trace_pf$v.txt 2>&1; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02p/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: round(v["avg_launch_ms"]*1e3,2) for n,v in k.items()}, round(d["passes_ms"]["tally"]*1e3,1))
PY
head -12 $O/trace_pf0.txt; head -12 $O/trace_pf3.txt
