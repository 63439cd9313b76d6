cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05n; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -x -q -k "adaln_qkv" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -3 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
AQKV_STAMPS=1 timeout -k 10 120 python tools/aqkv_probe.py 0 2>&1 | tail -14 | cut -c1-90 | tee -a $O/aqkv_probe.txt
