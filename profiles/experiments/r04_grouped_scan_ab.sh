O=gpurun_out/r04b4; mkdir -p $O
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gibbs" > $O/tests.txt 2>&1; rc=$?; tail -3 $O/tests.txt
[ $rc = 0 ] || exit 1
for v in "" _x0 _x1 _x2 _x3; do
  FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip$v.so timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/bench$v.json 2> $O/bench$v.err || exit 1
  python3 - $O/bench$v.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
k=d['kernels']
print("lib%-4s ms/sweep %.4f  samples/s %.0f  r %.1f us  f %.1f us" % (sys.argv[2], d['ms_per_step'], d['value'], k['gibbs_r_pipe_kernel']['avg_launch_ms']*1e3, k['gibbs_f_pair_kernel']['avg_launch_ms']*1e3))
PY
done
