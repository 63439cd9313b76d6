#!/bin/bash
# A/B of attention.hip builds on the GPU box (-D flags).
cd "$(dirname "$0")/.."
out=gpurun_out/attn_ab; mkdir -p $out
run() { # name, flags
  touch sea_amd/csrc/attention.hip
  SEA_EXTRA_FLAGS="$2" python -m sea_amd.build > $out/build_$1.log 2>&1 || { echo "build $1 failed"; tail -5 $out/build_$1.log; return; }
  python tools/bench_ops.py attn > $out/$1.txt 2>&1
  echo "== $1"; grep -v amdgpu.ids $out/$1.txt
}
run w8_6 ""
run w8_5 "-DSEA_ATTN_WPE32=5"
touch sea_amd/csrc/attention.hip; python -m sea_amd.build > /dev/null 2>&1
python -m pytest tests/test_ops_gpu.py tests/test_dropout_gpu.py tests/test_bwd_ops_gpu.py -x -q -k "attention" 2>&1 | tail -3
python -m pytest tests/test_model_gpu.py tests/test_train_gpu.py tests/test_encode_gpu.py tests/test_kv_fast_gpu.py -x -q 2>&1 | tail -5
python bench.py --mode rollout --no-cpu-baseline > $out/roll.json 2> $out/roll.err; python bench.py --mode train --steps 30 > $out/train.json 2> $out/train.err
