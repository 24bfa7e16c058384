# round 2, call r: timeline of the pipelined r pass
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02r; mkdir -p $O
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/trace_pipe.txt 2>&1; rc=$?
cat $O/trace_pipe.txt
