#!/bin/bash
# Is the cfg2 bf16 forward bit-identical run to run?  Under each plan switch (tools/graph_vs_plain.py prints the comparison).
for sw in "X=0" "SEA_FUSE_MLP1=0" "SEA_FUSE_XTAIL=0" "SEA_FUSE_NORM=0" "SEA_FOLD_IB=0" "SEA_FUSE_MLP1=0 SEA_FUSE_XTAIL=0 SEA_FUSE_NORM=0 SEA_FOLD_IB=0"; do
  echo "== $sw"; env $sw python tools/graph_vs_plain.py 2>/dev/null | grep -E "plain vs plain|rows that differ"
done
