#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03ag_tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03ag_tests.log
cat gpurun_out/tie_margin_cfg3.json
