cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
timeout -k 10 120 python tools/chain_probe.py stamps 2 > $O/stamps2.txt 2>&1
grep -v "Warning\|nanmax\|amdgpu.ids" $O/stamps2.txt
