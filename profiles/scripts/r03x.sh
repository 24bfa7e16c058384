#!/bin/bash
mkdir -p gpurun_out
export FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so
for d in 1 2 3; do
for v in 0 1; do
echo "FCD_ABL_DIAG=$d FCD_R_DSPLIT=$v"
FCD_ABL_DIAG=$d FCD_R_DSPLIT=$v timeout -k 10 200 python3 profiles/trace_pipe.py 2>&1 | cut -c1-170 | sed -n 6,9p
done
done > gpurun_out/r03x.txt 2>&1
