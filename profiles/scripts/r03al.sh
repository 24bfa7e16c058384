#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03al; mkdir -p $O
for m in lag blk none; do
  if [ $m = lag ]; then A="--force-pg --mstep-lag 1"; elif [ $m = blk ]; then A="--force-pg"; else A=""; fi
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/$m -o t -- python3 bench.py --steps 40 --warmup 5 $A --no-cpu-baseline --no-vb --no-corr > $O/$m.json 2> $O/$m.err || exit 1
done
python3 - <<'PY'
import csv, glob
for m in ("none", "blk", "lag"):
    f = glob.glob("gpurun_out/r03al/%s/**/*kernel_trace.csv" % m, recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    # the timed region: the last 30 f-pass kernels before the profiled second pass... take kernels 60%..80% of the trace
    names = [r["Kernel_Name"] for r in rows]
    idx = [i for i, n in enumerate(names) if "gibbs_f_pair_kernel" in n]
    a, b = idx[len(idx) // 3], idx[len(idx) // 3 + 8]
    print("==", m)
    prev_end = None
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gap = (s - prev_end) / 1e3 if prev_end else 0.0
        print("   %-46s start+%8.1f dur %7.1f gap %6.1f  q%s" % (r["Kernel_Name"][:46], (s - int(rows[a]["Start_Timestamp"])) / 1e3, (e - s) / 1e3, gap, r.get("Queue_Id", "")))
        prev_end = max(prev_end or 0, e)
PY
