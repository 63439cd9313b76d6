cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
timeout -k 10 120 python tools/ws_probe.py 256 > $O/probe256.txt 2>&1; grep "ws probe:" $O/probe256.txt
timeout -k 10 120 python tools/ws_probe.py 512 > $O/probe512.txt 2>&1; grep "ws probe:" $O/probe512.txt; grep "ws probe it 10" $O/probe512.txt
