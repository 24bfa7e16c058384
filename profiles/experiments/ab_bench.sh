# bench of the product library, 300 sweeps, three times (session noise): r pass / sweep
O=gpurun_out/ab; mkdir -p $O
for i in 1 2 3; do
  timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/b.json 2> $O/b.err || exit 1
  python3 - $O/b.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print("ms/sweep %.4f  samples/s %.0f  r %.1f us  f %.1f us  pack %.1f" % (d['ms_per_step'], d['value'], k['gibbs_r_pipe_kernel']['avg_launch_ms']*1e3, k['gibbs_f_pair_kernel']['avg_launch_ms']*1e3, k['pack_f_kernel']['avg_launch_ms']*1e3))
PY
done
