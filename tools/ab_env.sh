#!/bin/bash
# same-box A/B of an environment knob (the A/B switches are honoured under SRK_DEBUG=1 only): tools/ab_env.sh VAR model batch [reps]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=$1; M=$2; B=$3; R=${4:-3}
for r in $(seq $R); do
  for v in 0 1; do
    env SRK_DEBUG=1 $VAR=$v timeout 900 python3 bench.py --model $M --batch $B --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', '$M', 'b$B', d['value'], d['ms_per_step'])"
  done
done
