# round 3, call e: pipe2 with per-wave announce; roles alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 300 python3 profiles/dbg_r.py 2>&1 | grep -v amdgpu.ids | grep -c "mismatches 0 of"
B="python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-vb --no-corr"
timeout -k 10 300 $B > $O/pipe2.json 2> $O/pipe2.err; echo rc=$?
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r03e/pipe2.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],4), round(d["value"]), {k:round(v["avg_launch_ms"]*1e3,1) for k,v in d.get("kernels",{}).items()})
PY
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/ablate_pipe2.py > $O/ablate_pipe2.txt 2>&1; echo rc=$?
grep -v amdgpu.ids $O/ablate_pipe2.txt
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/trace_pipe2.txt 2>&1; echo rc=$?
grep -v amdgpu.ids $O/trace_pipe2.txt | cut -c1-150 | head -12; tail -1 $O/trace_pipe2.txt | cut -c1-200
