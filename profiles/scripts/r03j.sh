# round 3, call j: SQ counters of the pipelined pass (default build): full and the panel role alone
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03j; mkdir -p $O
for d in 5; do
  FCD_R_DBG=$d timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_a$d -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/pmc_a$d.log 2>&1; echo rc=$?
  FCD_R_DBG=$d timeout -k 10 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $O/pmc_b$d -o p -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-vb --no-corr > $O/pmc_b$d.log 2>&1; echo rc=$?
done
python3 - <<'PY'
import pandas as pd, glob
for d in (5,):
    for ab in "ab":
        fs=glob.glob("gpurun_out/r03j/pmc_%s%d/**/*counter_collection.csv"%(ab,d), recursive=True)
        if not fs: print("no csv", ab, d); continue
        df=pd.read_csv(fs[0])
        sub=df[df["Kernel_Name"].str.contains("gibbs_r_pipe", regex=False)]
        n=sub["Dispatch_Id"].nunique()
        avg=sub.groupby("Counter_Name")["Counter_Value"].sum()/n
        print("dbg=%d pass %s (%d launches):"%(d,ab,n), {k:int(v) for k,v in avg.items()})
PY
rm -rf $O/pmc_*/
