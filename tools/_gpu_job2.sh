cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_ops_gpu.py -x -q -k "b8 or B8 or batch or full_size or adaln_qkv or cfg2" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -3 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
for P in "" "front=0"; do
SEA_PLAN=$P python bench.py --mode rollout --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/rollout_b8.json 2> $O/b8.err
python -c "
import json
d=json.loads(open('$O/rollout_b8.json').read().strip().splitlines()[-1]); print('B=8 [$P]', d['ms_per_step']); print(d.get('launch_breakdown_ms') or d['rollout']['launch_breakdown_ms'])"
done
