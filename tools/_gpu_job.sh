cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
for PL in "" "silu=0" "silu=0,lanes=none"; do
echo "== SEA_PLAN=$PL"
SEA_PLAN=$PL timeout -k 10 300 python bench.py --mode rollout --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/b8.json 2> $O/b8.err; python -c "
import json
d=json.loads(open('$O/b8.json').read().strip().splitlines()[-1]); print('B=8 fwd ms', d['ms_per_step'], {k:v for k,v in d['rollout']['launch_breakdown_ms'].items() if 'adaln' in k or 'silu' in k})"
done
