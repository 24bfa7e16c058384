# round 2, call zd: step-per-launch form with the lean in-order scan (byte-form r words, LDS-space addresses); prefetch depth builds
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zd; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
if [ $rc -ne 0 ]; then echo "tests failed: stopping"; exit 1; fi
for lib in libfcdiff_hip libfcdiff_hip_s43 libfcdiff_hip_s44 libfcdiff_hip_s54 libfcdiff_hip libfcdiff_hip_s43 libfcdiff_hip_s44; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_${lib}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
FCD_R_PATH=2 timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_pipe.json 2>> $O/bench.err
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zd/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_r.py > $O/trace_r.txt 2>&1; head -12 $O/trace_r.txt
