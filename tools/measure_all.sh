#!/bin/bash
# Round measurement set (run on the MI355X box from the repository root): bench lines, rocprofv3 kernel stats / traces, PMC passes.
#   tools/measure_all.sh [tag] [stage ...]   -> gpurun_out/<tag>/   (default tag: final; stages: bench traces pmcf pmct, default all four);
#   tools/collect_profiles.py <tag> rNN copies the judged files into profiles/
# Counter passes are processes of their own with --kernel-trace only (never a hip / hsa / sys trace beside --pmc), the program directly behind `--`.
# A trace step fails on ANY non-zero exit of rocprofv3 (round 3 tolerated a SIGSEGV inside exit() of the KV leg; its cause — ROCr tearing down the cooperative
# queue under the tool's queue interception — is recorded in profiles/failures/r04_rocprof_kv_exit_stack.txt and avoided in kvstep.hip).
set -o pipefail
O=gpurun_out/${1:-final}
shift
STAGES=${*:-bench traces pmcf pmct}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
has() { [[ " $STAGES " == *" $1 "* ]]; }
prof() {   # prof <dir> <log> <program...>: kernel trace + per-kernel summary; succeeds when the summary exists
  local d=$1 log=$2; shift 2
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$d -o run -- "$@" > $O/$log 2> $O/$d.err
  local rc=$?
  find $O/$d -name "*kernel_stats.csv" | grep -q . || { echo "rocprofv3 $d: no summary (exit $rc)"; return 1; }
  if [ $rc -ne 0 ]; then echo "rocprofv3 $d: EXIT $rc (signal $((rc > 128 ? rc - 128 : 0)))"; cp $O/$d.err $O/${d}_rocprof_exit_$rc.err; return 1; fi
  return 0
}
if has bench; then
python bench.py > $O/bench_n1.json 2> $O/bench_n1.err || exit 1
python bench.py --mode rollout --steps 200 --warmup 10 --no-cpu-baseline --graph > $O/bench_rollout_graph.json 2>/dev/null || exit 1
python bench.py --mode rollout --batch 8 --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_rollout_b8.json 2>/dev/null || exit 1
python bench.py --mode decode --steps 10 --warmup 3 > $O/bench_decode.json 2>/dev/null || exit 1
python bench.py --mode encode --steps 10 --warmup 3 > $O/bench_encode.json 2>/dev/null || exit 1
echo "bench lines done"
fi
if has traces; then
prof prof_fwd prof_fwd.json python3 bench.py --mode rollout --steps 50 --warmup 10 --no-cpu-baseline --no-graph || exit 1
prof prof_train prof_train.json python3 bench.py --mode train --steps 10 --warmup 3 || exit 1
prof prof_kv prof_kv.json python3 bench.py --mode kv --steps 3 --warmup 1 || exit 1
prof prof_kvs prof_kvs.log python3 tools/kv_shipped.py 100 || exit 1
python tools/kv_shipped.py 100 > $O/kv_shipped_widths.txt 2>/dev/null || exit 1
python tools/train_launch_table.py 8 > $O/train_launch_table.txt 2>/dev/null || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_tt -o run -- python3 tools/train_trace.py run > $O/prof_tt.log 2>&1 || exit 1
python tools/train_trace.py show $(find $O/prof_tt -name "*kernel_trace.csv" | head -1) > $O/train_step_trace.txt 2>&1 || exit 1
echo "traces done"
fi
if has pmcf; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o run -- python3 tools/pmc_forward.py > $O/pmc_f.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o run -- python3 tools/pmc_forward.py > $O/pmc_w.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_u -o run -- python3 tools/pmc_forward.py > $O/pmc_u.log 2>&1 || exit 1
cp gpurun_out/pmc_forward_names.json $O/
python tools/pmc_traffic.py $O/pmc_f $O/pmc_w $O/pmc_forward_names.json $O/forward_cfg2_pmc_traffic.json > $O/pmc_traffic.log 2>&1 || exit 1
python tools/pmc_util.py $O/pmc_u $O/pmc_forward_names.json $O/forward_cfg2_pmc_utilisation.json > $O/pmc_util.log 2>&1 || exit 1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_graph -o run -- python3 bench.py --mode rollout --steps 40 --warmup 5 --no-cpu-baseline --graph > $O/prof_graph.json 2> $O/prof_graph.err
NL=$(python -c "import json; print(len(json.load(open('$O/pmc_forward_names.json'))['names']))")
python tools/trace_breakdown.py $(find $O/prof_graph -name "*kernel_trace.csv" | head -1) $NL $O/pmc_forward_names.json > $O/forward_cfg2_launch_breakdown.txt 2>&1 || exit 1
echo "forward pmc done"
fi
if has pmct; then
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_tf -o run -- python3 tools/train_trace.py run > $O/pmc_tf.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_tw -o run -- python3 tools/train_trace.py run > $O/pmc_tw.log 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $O/pmc_tu -o run -- python3 tools/train_trace.py run > $O/pmc_tu.log 2>&1 || exit 1
python tools/pmc_train.py $O/pmc_tf $O/pmc_tw $O/pmc_tu $O/train_cfg3_pmc_traffic.json $O/train_cfg3_pmc_utilisation.json > $O/pmc_train.log 2>&1 || exit 1
echo "train pmc done"
fi
# the raw counter / trace directories stay on the box side of the merge limit: keep the folded files and the per-kernel summaries only
for d in pmc_f pmc_w pmc_u pmc_tf pmc_tw pmc_tu prof_graph prof_tt; do rm -rf $O/$d; done
find $O -name "*kernel_trace.csv" -size +8M -delete
echo "all done"
