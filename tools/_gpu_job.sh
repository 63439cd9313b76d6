cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r05r; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_ptrcheck_gpu.py -x -q -k "adaln_qkv or mlp_block or mlp_fc or cfg2 or optional or ptrcheck or corrupted or audit or golden" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -6 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
for P in "" "front3=0" "front=0"; do
SEA_PLAN=$P timeout -k 10 120 python tools/chain_probe.py replay 100 > $O/replay.txt 2>&1; echo "[$P] $(tail -1 $O/replay.txt | cut -c1-45)"
done
python bench.py --mode rollout --steps 20 --warmup 5 --no-cpu-baseline > $O/rollout.json 2> $O/rollout.err
python -c "
import json
d=json.loads(open('$O/rollout.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step']); print(d.get('launch_breakdown_ms') or d['rollout']['launch_breakdown_ms']); print(d['roofline'])"
