# round 2, call zp: pipelined r pass, f words of the in-order scan asked for 2 / 3 rows ahead (builds PF_E/PF_F = 3/3, 2/3, 3/4)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zp; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
for lib in libfcdiff_hip_pf33 libfcdiff_hip_pf23 libfcdiff_hip_pf34; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "pipelined" > $O/gpu_tests_$lib.txt 2>&1; rc=$?; tail -1 $O/gpu_tests_$lib.txt; stop_if_killed $rc
  if [ $rc -ne 0 ]; then echo "pipelined tests failed: stopping"; exit 1; fi
done
for lib in libfcdiff_hip libfcdiff_hip_pf33 libfcdiff_hip_pf23 libfcdiff_hip_pf34 libfcdiff_hip libfcdiff_hip_pf33 libfcdiff_hip_pf23 libfcdiff_hip_pf34; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_${lib}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zp/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
