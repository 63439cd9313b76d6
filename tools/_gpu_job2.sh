cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06e; mkdir -p $O
for P in "" "rider_caps=0:0:0"; do
SEA_PLAN=$P timeout -k 10 120 python tools/chain_probe.py replay 100 > $O/replay.txt 2>&1; echo "[$P] $(tail -1 $O/replay.txt | cut -c1-45)"
done
SEA_PLAN=rider_caps=0:0:0 python bench.py --mode rollout --steps 20 --warmup 5 --no-cpu-baseline > $O/rollout.json 2> $O/rollout.err
python -c "
import json
d=json.loads(open('$O/rollout.json').read().strip().splitlines()[-1]); print('bench', d['ms_per_step']); print(d.get('launch_breakdown_ms') or d['rollout']['launch_breakdown_ms'])"
