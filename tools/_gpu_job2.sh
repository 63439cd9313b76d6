cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
for t in "blk_perxcd=1" "blk_perxcd=0"; do for m in 4048 16192; do SEA_TUNE=$t timeout -k 10 120 python tools/mlp_probe.py $m 2>&1 | tail -1 | cut -c1-70 | tee -a $O/mlp_probe.txt; done; done
