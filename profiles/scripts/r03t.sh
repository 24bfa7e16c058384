#!/bin/bash
mkdir -p gpurun_out
export FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so
FCD_ABL_PANEL=2 timeout -k 10 200 python3 profiles/trace_pipe.py > gpurun_out/r03t_trace_nopanelterms.txt 2>&1 || exit 1
FCD_ABL_PANEL=3 timeout -k 10 200 python3 profiles/trace_pipe.py > gpurun_out/r03t_trace_nopanelbuild.txt 2>&1 || exit 1
timeout -k 10 300 python3 profiles/ablate_pipe.py > gpurun_out/r03t_ablate.txt 2>&1
