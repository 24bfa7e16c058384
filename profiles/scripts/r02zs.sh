# round 2, call zs: K_lik with non-temporal stores of the table (build libfcdiff_hip_liknt.so) against the default
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zs; mkdir -p $O
for lib in libfcdiff_hip libfcdiff_hip_liknt libfcdiff_hip libfcdiff_hip_liknt; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_${lib}_$RANDOM.json 2>> $O/bench.err
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --nreg 400 --subjects 500 --steps 2 --warmup 1 --settle 0 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_${lib}_$RANDOM.json 2>> $O/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zs/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["lik_tables"]["avg_launch_ms"]*1e3,2), round(d["lik_tables"]["frac"],3))
PY
