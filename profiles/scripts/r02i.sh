# round 2, call i: phase ablation of the current build (ablation library), tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02i; mkdir -p $O
stop_if_killed() { if [ "$1" = "124" ] || [ "$1" = "137" ]; then echo "step killed at its limit (rc $1): stopping"; exit 1; fi; }
timeout -k 10 900 python3 -m pytest tests -m gpu -q -x > $O/gpu_tests.txt 2>&1; rc=$?; tail -3 $O/gpu_tests.txt; stop_if_killed $rc
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/ablate.py > $O/ablate.txt 2>&1; rc=$?; stop_if_killed $rc; cat $O/ablate.txt
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_r.py > $O/trace_r.txt 2>&1; rc=$?; stop_if_killed $rc; head -12 $O/trace_r.txt
