cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06x; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py -x -q -k "gemm or cfg3 or gradients or golden" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -5 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
for TN in "gemm_ws=0" ""; do
echo "== SEA_TUNE=$TN"
SEA_TUNE=$TN timeout -k 10 300 python bench.py --mode train --steps 30 --no-cpu-baseline > $O/train_$TN.json 2> $O/train.err; python -c "
import json
d=json.loads(open('$O/train_$TN.json').read().strip().splitlines()[-1]); t=d.get('train',d); print('train ms', t.get('ms_per_step'), t['top_launches_ms'])"
done
