"""Per-launch device durations of one plan replay from a rocprofv3 --kernel-trace CSV: finds the longest stretch of the trace in which the
kernel-name sequence repeats with period L (the plan's launch count) and averages every position over the periods.
    python tools/trace_breakdown.py <kernel_trace.csv> <L> [names.json]"""
import csv
import json
import re
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_Z\d+([a-z_0-9]+?)I", n)
    return (m.group(1) if m else n.split("<")[0].split("(")[0])[:28]


def main():
    path, L = sys.argv[1], int(sys.argv[2])
    labels = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
    if isinstance(labels, dict):
        labels = labels["names"]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    start = [int(r["Start_Timestamp"]) for r in rows]
    end = [int(r["End_Timestamp"]) for r in rows]
    n = len(names)
    same = [i + L < n and names[i] == names[i + L] for i in range(n)]
    best, cur, s0 = (0, 0), 0, 0
    for i in range(n):
        if same[i]:
            if cur == 0:
                s0 = i
            cur += 1
            if cur > best[0]:
                best = (cur, s0)
        else:
            cur = 0
    length, s0 = best
    periods = length // L
    if periods < 2:
        print("no periodic stretch found", length)
        return
    s0 += L  # skip the first period (warm-up effects)
    periods -= 1
    tot = [0.0] * L
    gap = [0.0] * L
    for p in range(periods):
        for k in range(L):
            i = s0 + p * L + k
            tot[k] += dur[i]
            gap[k] += start[i] - end[i - 1]
    print(f"{periods} periods of {L} launches from trace index {s0}")
    ssum = gsum = 0.0
    for k in range(L):
        d, g = tot[k] / periods / 1000, gap[k] / periods / 1000
        ssum += d
        gsum += g
        lab = labels[k] if labels else ""
        print(f"{k:3d} {short(names[s0 + k]):30s} {lab:28s} {d:8.2f} us   gap before {g:6.2f}")
    print(f"sum of durations {ssum:.1f} us, sum of gaps {gsum:.1f} us, period {ssum + gsum:.1f} us")


main()
