# A/B of the half-block hand-over build under the placement knobs (bench, 300 sweeps each)
O=gpurun_out/r04e3; mkdir -p $O
run() { env $1 timeout -k 10 200 python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline --no-vb --no-corr > $O/b.json 2> $O/b.err || exit 1
  python3 - $O/b.json "$1" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); k=d['kernels']
print("%-34s ms/sweep %.4f  r %.1f us" % (sys.argv[2], d['ms_per_step'], k['gibbs_r_pipe_kernel']['avg_launch_ms']*1e3))
PY
}
run "FCD_X=0"
run "FCD_R_DSPLIT=1"
run "FCD_R_DSPLIT=1 FCD_R_NOPAD=1"
run "FCD_R_NOPAD=1"
run "FCD_X=0"
