#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03af; mkdir -p $O
export FCDIFF_HIP_LIB=profiles/var_corr4.so
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o corr -- python3 profiles/corr_only.py 20 > $O/run.log 2>&1 || exit 1
f=$(find $O/k -name "*kernel_stats.csv" | head -1); head -3 "$f" | cut -c1-200
