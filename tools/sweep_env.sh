#!/bin/bash
# tools/sweep_env.sh VAR "v1 v2 ..." [bench args...]: the training step time (bench.py --mode train) for each value of an environment switch (tuning aid)
VAR=$1; VALS=$2; shift 2
for v in $VALS; do
  export $VAR=$v
  python bench.py --mode train --steps 25 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(os.environ['$VAR'], d['train']['ms_per_step'])"
done
