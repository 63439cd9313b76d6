cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06o; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_dropout_gpu.py -x -q -k "attention" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -6 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
for X in 0 1; do
echo "== attn_paired=$X"; SEA_TUNE=attn_paired=$X timeout -k 10 300 python bench.py --mode rollout --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/b8_$X.json 2> $O/b8_$X.err; python -c "
import json
d=json.loads(open('$O/b8_$X.json').read().strip().splitlines()[-1]); print('B=8 fwd ms', d['ms_per_step'], {k:v for k,v in d['launch_breakdown_ms'].items() if 'attention' in k})"
SEA_TUNE=attn_paired=$X timeout -k 10 300 python bench.py --mode train --steps 30 --no-cpu-baseline > $O/train_$X.json 2> $O/train_$X.err; python -c "
import json
d=json.loads(open('$O/train_$X.json').read().strip().splitlines()[-1]); t=d.get('train',d); print('train ms', t.get('ms_per_step'), t['top_launches_ms'].get('self.attention'))"
done
