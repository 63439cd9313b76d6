cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 python tools/write_probe.py 2>&1 | grep MB
