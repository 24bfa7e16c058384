# round 3, call c: timeline of the pipelined pass with tagged hand-overs (diagnostic build)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/trace_pipe2.txt 2>&1; echo rc=$?
grep -v amdgpu.ids $O/trace_pipe2.txt
