cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
for T in "gemm_tile=128" "gemm_tile=128,gemm_n_major=0" "gemm_tile=64,gemm_dma_min_k=100000"; do
SEA_TUNE=$T python bench.py --mode shipped --steps 10 --warmup 3 > $O/shipped_t.json 2> $O/shipped_t.err
python - "$T" <<'PY'
import json,sys
d=json.loads(open("gpurun_out/r04g/shipped_t.json").read().strip().splitlines()[-1])
sh=d.get("shipped",d)
for k,v in sh.items():
    if isinstance(v,dict): print(sys.argv[1], k, round(v["forward_ms"],4), round(v["train_ms_per_step"],4), v["top_launches_ms"])
PY
done
