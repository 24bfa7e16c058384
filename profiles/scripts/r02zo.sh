# round 2, call zo: the diagnostic files of the pipelined r pass (strand ablation, timeline, row stamps) with the final build
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02zo; mkdir -p $O
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/ablate_pipe.py > $O/r02_ablate_pipe.txt 2>&1; cat $O/r02_ablate_pipe.txt
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/r02_trace_pipe.txt 2>&1; tail -8 $O/r02_trace_pipe.txt
FCD_TRACE_ROW=1 FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/trace_pipe.py > $O/r02_trace_pipe_row_stamps.txt 2>&1; tail -9 $O/r02_trace_pipe_row_stamps.txt
