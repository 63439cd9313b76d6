cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
python bench.py --mode rollout --steps 20 --warmup 5 --no-cpu-baseline > $O/rollout.json 2> $O/rollout.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04l/rollout.json").read().strip().splitlines()[-1])
print("bench ms_per_step", d["ms_per_step"], "launches", len(d["launch_breakdown_ms"]) if "launch_breakdown_ms" in d else d.get("rollout",{}).get("n_launches"))
print(d.get("launch_breakdown_ms") or d["rollout"]["launch_breakdown_ms"])
PY
python bench.py --mode rollout --steps 200 --warmup 10 --no-cpu-baseline > $O/rollout200.json 2> $O/rollout200.err
python -c "
import json
d=json.loads(open('gpurun_out/r04l/rollout200.json').read().strip().splitlines()[-1]); print('200 steps', d['ms_per_step'])"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --deselect tests/test_train_gpu.py::test_cfg3_own_batch_gradients_match_oracle > $O/test_gpu.log 2>&1; echo "gpu rc=$?" >> $O/test_gpu.log; tail -5 $O/test_gpu.log
