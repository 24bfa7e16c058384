#!/bin/bash
# timeline of the pipelined r pass (ablation build), both layouts of the in-order role
mkdir -p gpurun_out
for v in 0 1; do
FCD_R_DSPLIT=$v FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 200 python3 profiles/trace_pipe.py > gpurun_out/r03r_trace_$v.txt 2>&1 || exit 1
done
