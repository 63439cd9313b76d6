cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_ptrcheck_gpu.py tests/test_ops_gpu.py -x -q -k "optional or rider or forward_fp32 or ablation or ptrcheck or corrupted or silu or shipped or cfg2" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -4 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
for P in "" "projnorm=0,fold_ib_gen=0"; do
SEA_PLAN=$P python bench.py --mode rollout --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/rollout_b8.json 2>/dev/null
python -c "
import json
d=json.loads(open('gpurun_out/r04n/rollout_b8.json').read().strip().splitlines()[-1]); print('B=8 [$P]', d['ms_per_step']); print(d.get('launch_breakdown_ms') or d['rollout']['launch_breakdown_ms'])"
done
