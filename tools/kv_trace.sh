#!/bin/bash
# Kernel trace of the KV-cache rollout (one rollout of 2024 steps after one warm-up): per-launch device time of a step vs the host-side period
O=gpurun_out/kv_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $O -o run -- python3 bench.py --mode kv --steps 1 --warmup 1 > $O/bench.json 2> $O/err.log || exit 1
python3 - <<'P'
import csv, collections
rows = sorted(csv.DictReader(open("gpurun_out/kv_trace/run_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]            # the timed rollout
n = len(rows)
dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print("kernels", n, "device busy %.1f ms" % (dur / 1e6), "span %.1f ms" % (span / 1e6), "avg kernel %.2f us" % (dur / n / 1e3), "avg period %.2f us" % (span / n / 1e3))
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r["Kernel_Name"][:60]; agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s n=%6d avg %.2f us total %.1f ms" % (k, c, t / c / 1e3, t / 1e6))
P
