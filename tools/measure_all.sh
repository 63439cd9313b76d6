#!/bin/bash
# Round measurement set (run on the MI355X box from the repository root): bench lines, rocprofv3 kernel stats / traces, PMC passes.
#   tools/measure_all.sh [tag]      -> gpurun_out/<tag>/   (default tag: final)
set -o pipefail
O=gpurun_out/${1:-final}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
python bench.py --mode rollout --steps 200 --warmup 10 --no-cpu-baseline --graph > $O/bench_rollout_graph.json 2>/dev/null || exit 1
python bench.py --mode rollout --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_rollout_b8.json 2>/dev/null || exit 1
python bench.py --mode decode --steps 10 --warmup 3 > $O/bench_decode.json 2>/dev/null || exit 1
python bench.py --mode encode --steps 10 --warmup 3 > $O/bench_encode.json 2>/dev/null || exit 1
echo "bench lines done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_fwd -o run -- python3 bench.py --mode rollout --steps 50 --warmup 10 --no-cpu-baseline --no-graph > $O/prof_fwd.json 2> $O/prof_fwd.err || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_graph -o run -- python3 bench.py --mode rollout --steps 40 --warmup 5 --no-cpu-baseline --graph > $O/prof_graph.json 2> $O/prof_graph.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o run -- python3 bench.py --mode train --steps 10 --warmup 3 > $O/prof_train.json 2> $O/prof_train.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kv -o run -- python3 bench.py --mode kv --steps 3 --warmup 1 > $O/prof_kv.json 2> $O/prof_kv.err || exit 1
SEA_TUNE=kv_persist=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kv7 -o run -- python3 bench.py --mode kv --steps 3 --warmup 1 > $O/prof_kv7.json 2> $O/prof_kv7.err || exit 1
python tools/kv_persist_timeline.py 2024 > $O/kv_timeline.txt 2>/dev/null || exit 1
echo "traces done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o run -- python3 tools/pmc_forward.py > $O/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o run -- python3 tools/pmc_forward.py > $O/pmc_w.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_u -o run -- python3 tools/pmc_forward.py > $O/pmc_u.log 2>&1 || exit 1
cp gpurun_out/pmc_forward_names.json $O/
echo "pmc done"
