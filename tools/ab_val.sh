#!/bin/bash
# same-box A/B of an environment VALUE: tools/ab_val.sh VAR "v1 v2 ..." model batch
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=$1; VALS=$2; M=$3; B=$4
for r in 1 2; do
  for v in $VALS; do
    env $VAR=$v timeout 900 python3 bench.py --model $M --batch $B --steps 20 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v', '$M', 'b$B', d['value'], d['ms_per_step'])"
  done
done
