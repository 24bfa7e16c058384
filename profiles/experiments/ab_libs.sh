# bench of several builds of the library (FCDIFF_HIP_LIB), 300 sweeps each
O=gpurun_out/ab; mkdir -p $O
for v in "$@"; do
  FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip$v.so timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python3 - $O/b.json "$v" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print("lib%-6s ms/sweep %.4f  samples/s %.0f  r %.1f us  f %.1f us  pack %.1f" % (sys.argv[2], d['ms_per_step'], d['value'], k['gibbs_r_pipe_kernel']['avg_launch_ms']*1e3, k['gibbs_f_pair_kernel']['avg_launch_ms']*1e3, k['pack_f_kernel']['avg_launch_ms']*1e3))
PY
done
