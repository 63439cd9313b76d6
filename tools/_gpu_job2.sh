cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -x -q -k "mlp_block or cfg2" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -3 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
for p in 1 2 0; do SEA_TUNE=blk_probe=$p timeout -k 10 120 python tools/mlp_probe.py 2>&1 | tail -1 | cut -c1-60 | tee -a $O/mlp_probe.txt; done
timeout -k 10 120 python tools/chain_probe.py replay 100 2>&1 | tail -1 | cut -c1-45
