# round 2, call ze: in-order scan, the four waves of a SIMD started a fraction of a row apart (builds 2 / 4 / 8 sleep units per slot)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02ze; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
for lib in libfcdiff_hip libfcdiff_hip_ds2 libfcdiff_hip_ds4 libfcdiff_hip_ds8 libfcdiff_hip; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_${lib}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02ze/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
