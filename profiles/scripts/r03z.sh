#!/bin/bash
mkdir -p gpurun_out
export FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so
for v in 0 1; do
FCD_R_DSPLIT=$v timeout -k 10 200 python3 profiles/trace_pipe.py > gpurun_out/r03z_trace_$v.txt 2>&1 || exit 1
done
