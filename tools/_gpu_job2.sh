cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06h; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "adaln_qkv or cfg2" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -3 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
AQKV_STAMPS=1 timeout -k 10 120 python tools/chain_probe.py replay 100 2>&1 | tail -13 | cut -c1-100 | tee $O/stamps.txt
