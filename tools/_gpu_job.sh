cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06d; mkdir -p $O
for X in 0 3; do
export SEA_TUNE=attnb_mode=$X
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_$X -o run -- python3 tools/bench_attn_bwd.py > $O/pmc_$X.log 2>&1 || { tail -5 $O/pmc_$X.log; exit 1; }
echo "== mode $X"; python tools/pmc_attnb.py $O/pmc_$X | tee $O/traffic_$X.txt
rm -rf $O/pmc_$X
done
