cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "rider_plan" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -5 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
timeout -k 10 120 python tools/chain_probe.py replay 50 > $O/replay.txt 2>&1; tail -1 $O/replay.txt | cut -c1-60
bash tools/_gpu_job.sh
