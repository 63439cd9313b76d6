cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/gemm_census.py 8 2>&1 | grep "groups:" > gpurun_out/gemm_census.txt; cat gpurun_out/gemm_census.txt
