# round 2, call zt: K_lik: non-temporal stores (now default) and, in the second build, non-temporal loads of bt as well; K_lik tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zt; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -k "lps or lik or update_lps or vb or fit" > $O/gpu_tests_lik.txt 2>&1; tail -2 $O/gpu_tests_lik.txt
for lib in libfcdiff_hip libfcdiff_hip_liknt libfcdiff_hip libfcdiff_hip_liknt; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --settle 0 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg3_${lib}_$RANDOM.json 2>> $O/bench.err
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --nreg 400 --subjects 500 --steps 2 --warmup 1 --settle 0 --no-cpu-baseline --no-vb --no-corr > $O/bench_cfg5_${lib}_$RANDOM.json 2>> $O/bench.err
done
for lib in libfcdiff_hip libfcdiff_hip_packnt libfcdiff_hip libfcdiff_hip_packnt; do
  FCDIFF_HIP_LIB=fcdiff_amd/$lib.so timeout -k 10 300 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/sweep_cfg3_${lib}_$RANDOM.json 2>> $O/bench.err
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zt/sweep_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); k=d["kernels"]
    print(f, round(d["ms_per_step"],4), {n: (round(v["avg_launch_ms"]*1e3,2), v["launches"]) for n,v in k.items()})
PY
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02zt/bench_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["lik_tables"]["avg_launch_ms"]*1e3,2), round(d["lik_tables"]["frac"],3))
PY
