#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ae; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o corr -- python3 profiles/corr_only.py 20 > $O/run.log 2>&1 || exit 1
f=$(find $O/k -name "*kernel_stats.csv" | head -1); head -6 "$f" | cut -c1-160
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F64 --kernel-trace --output-format csv -d $O/p -o corr -- python3 profiles/corr_only.py 5 > $O/pmc.log 2>&1 || { tail -5 $O/pmc.log; exit 1; }
f=$(find $O/p -name "*counter_collection.csv" | head -1); python3 - "$f" <<'PY'
import csv,sys,collections
d=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:60]; d[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
for k,v in d.items(): print(k, dict(v))
PY
