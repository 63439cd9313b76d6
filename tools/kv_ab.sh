run() { env "$@" python bench.py --mode kv --steps 20 --warmup 3 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'], 4))"; }
for i in 1 2; do
run X=0
run SEA_FUSED=1
run SEA_FUSE_TAIL=1 SEA_FUSE_FINAL=1
run SEA_FUSE_OPROJ=1
run SEA_FUSE_KV=1
run SEA_FUSE_TAIL=1 SEA_FUSE_FINAL=1 SEA_FUSE_OPROJ=1 SEA_FUSE_KV=1
run SEA_FUSE_MLP1=1
run SEA_FUSE_SILU=1
done
