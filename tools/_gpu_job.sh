cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_kv_fast_gpu.py tests/test_model_gpu.py tests/test_fewrows_gpu.py -k "kv or rollout or persist or fewrows" -x -q > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -4 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
timeout -k 10 300 python bench.py --mode kv --no-cpu-baseline > $O/kv.json 2> $O/kv.err; python -c "
import json
d=json.loads(open('$O/kv.json').read().strip().splitlines()[-1]); k=d['kv']
for n,v in k.items(): print(n, v['ms_per_step'])"
