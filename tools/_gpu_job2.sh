cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_train_gpu.py -x -q -k "gemm or cfg3 or width" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -3 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
python bench.py --mode shipped --no-cpu-baseline > $O/shipped.json 2> $O/shipped.err
python -c "
import json
d=json.loads(open('$O/shipped.json').read().strip().splitlines()[-1])
for k,v in d['shipped'].items(): print(k, v['forward_ms'], v['train_ms_per_step'])"
python bench.py --mode train --steps 15 --warmup 5 --no-cpu-baseline > $O/train.json 2> $O/train.err
python -c "
import json
d=json.loads(open('$O/train.json').read().strip().splitlines()[-1]); print('train', d['ms_per_step'])"
python bench.py --mode rollout --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/rollout_b8.json 2> $O/b8.err
python -c "
import json
d=json.loads(open('$O/rollout_b8.json').read().strip().splitlines()[-1]); print('B=8', d['ms_per_step'])"
