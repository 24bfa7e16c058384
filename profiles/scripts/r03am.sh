#!/bin/bash
# the GPU suite with other DEFAULT forms selected through the environment (knobs are read once, at fcd_ctx_create)
mkdir -p gpurun_out
for e in "FCD_R_DSPLIT=1" "FCD_R_PATH=3" "FCD_R_REFILL=1 FCD_CORR_FORM=1" "FCD_F_FORM=2"; do
  echo "== $e" >> gpurun_out/r03am.txt
  env $e timeout -k 10 600 python3 -m pytest tests -m gpu -q 2>&1 | tail -3 >> gpurun_out/r03am.txt
done
cat gpurun_out/r03am.txt
