cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06k; mkdir -p $O
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err ) 2> $O/time.txt; tail -3 $O/time.txt
python -c "
import json
d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['rollout']['ms_per_step_from_idle'], d['roofline']['frac'], d['train']['ms_per_step'])"
