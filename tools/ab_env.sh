#!/bin/bash
# Interleaved A/B of one environment switch on the same box: tools/ab_env.sh VAR "bench.py arguments" [rounds] [launch name to print from launch_breakdown_ms]
VAR=$1; ARGS=$2; R=${3:-3}; NAME=${4:-}
for i in $(seq $R); do for v in 1 0; do
  env $VAR=$v python bench.py $ARGS --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', '$ARGS', round(d['ms_per_step'], 4), (d.get('roofline') or {}).get('kernel'), (d.get('roofline') or {}).get('launch_ms'), (d.get('launch_breakdown_ms') or {}).get('$NAME'))"
done; done
