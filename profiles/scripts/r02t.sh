cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02t; mkdir -p $O
FCDIFF_HIP_LIB=fcdiff_amd/libfcdiff_hip_abl.so timeout -k 10 300 python3 profiles/ablate_pipe.py > $O/ablate_pipe.txt 2>&1; cat $O/ablate_pipe.txt
