#!/bin/bash
# Probe build of sea_mlp_fc1_ln_gelu: launch time with the MFMAs and / or the epilogue switched off (SEA_MLP_PROBE bit 0 / bit 1)
for pr in ${PROBES:-0 1 2 3}; do
  SEA_MLP_PROBE=$pr python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe $pr', d['launch_breakdown_ms'].get('mlp.fc1_ln_gelu'), d['roofline']['kernel'], d['roofline']['launch_ms'])"
done
