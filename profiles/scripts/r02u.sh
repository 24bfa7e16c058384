cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02u; mkdir -p $O
FCD_ABL_PANEL=2 FCD_ABL_DIAG=2 FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/trace_pipe_skeleton.txt 2>&1; cat $O/trace_pipe_skeleton.txt
