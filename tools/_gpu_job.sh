cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r07f; mkdir -p $O
timeout -k 10 1150 python -m pytest tests/ -x -q -m gpu > $O/test_all.log 2>&1; echo "rc=$?" >> $O/test_all.log; tail -4 $O/test_all.log
