#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ai; mkdir -p $O
for T in 128 600 1200 2400; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k$T -o corr -- python3 profiles/corr_only.py 20 $T > $O/run$T.log 2>&1 || exit 1
f=$(find $O/k$T -name "*kernel_stats.csv" | head -1); echo "T=$T"; grep "corr_" "$f" | cut -d, -f1,4 | cut -c1-60,120-
done
