#!/bin/bash
# default plan against the parallel-branch plans (SEA_PLAN_LANES=cond|all), same box
for i in 1 2; do for v in none cond all; do
  SEA_PLAN_LANES=$v python bench.py --no-cpu-baseline ${ARGS:-} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes=$v', round(d['ms_per_step'], 4))"
done; done
