#!/bin/bash
cd "$(dirname "$0")/.."
out=gpurun_out/attnb_ab; mkdir -p $out
run() { touch sea_amd/csrc/attention_bwd.hip; SEA_EXTRA_FLAGS="$2" python -m sea_amd.build > $out/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $out/build_$1.log; return; }
  python tools/bench_attn_bwd.py > $out/$1.txt 2>&1; echo "== $1"; grep -v amdgpu.ids $out/$1.txt; }
run w1 ""
run w5 "-DSEA_ATTNB_WPE=5"
run w6 "-DSEA_ATTNB_WPE=6"
run w8 "-DSEA_ATTNB_WPE=8"
touch sea_amd/csrc/attention_bwd.hip; python -m sea_amd.build > /dev/null 2>&1
