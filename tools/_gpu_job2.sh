cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06l; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py tests/test_train_gpu.py -x -q -k "splitk or multiphase or cfg5" > $O/test.log 2>&1; echo "rc=$?" >> $O/test.log; tail -3 $O/test.log
grep -q "rc=0" $O/test.log || exit 1
for P in "" "splitk=0"; do
SEA_PLAN=$P python bench.py --mode shipped --no-cpu-baseline > $O/shipped.json 2> $O/shipped.err
python -c "
import json
d=json.loads(open('$O/shipped.json').read().strip().splitlines()[-1])
v=d['shipped']['multiphase_flow']; print('[$P]', v['forward_ms'], v['train_ms_per_step'], v['top_launches_ms'])"
done
