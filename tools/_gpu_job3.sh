cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > $O/test_gpu.log 2>&1; echo "rc=$?" >> $O/test_gpu.log; tail -5 $O/test_gpu.log
