cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_parallel_gpu.py tests/test_bwd_ops_gpu.py -x -q -k "rccl or attention" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -6 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
SEA_DP_REHEARSE=1 timeout -k 10 600 python bench.py --mode train --no-cpu-baseline > $O/train_rehearse.json 2> $O/train_rehearse.err; echo "bench rc=$?"; tail -3 $O/train_rehearse.err; python -c "
import json
d=json.loads(open('$O/train_rehearse.json').read().strip().splitlines()[-1]); t=d.get('train',d)
print({k:t.get(k) for k in ('ms_per_step','allreduce_calls_per_step','allreduce_ms','allreduce_bytes','backend','single_gpu_ms_per_step','parameters_in_sync_after_run')})"
