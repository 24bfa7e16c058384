# round 3, call n: robustness tests (give-up reported, fallback form, foreign kernel beside the pipelined pass, RCCL group of one rank)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_dist_nccl.py tests/test_abi.py -x -q -k "give_up or falls_back or foreign or one_rank or pipelined or abi or symbols" > $O/tests.txt 2>&1; echo rc=$?
tail -15 $O/tests.txt
