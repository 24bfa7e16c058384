# round 2, call zc: cfg5, blocks per group of prefetched state words in the one-patient panel role (2, 3, 4, 6)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zc; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
for lib in libfcdiff_hip libfcdiff_hip_pg3 libfcdiff_hip_pg4 libfcdiff_hip_pg6 libfcdiff_hip; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 600 python3 bench.py --nreg 400 --subjects 500 --steps 10 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_${lib}_$RANDOM.json 2>> $O/bench.err; rc=$?; stop_if_killed $rc
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zc/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
