#!/bin/bash
# ON THE GPU BOX: same-box A/B of the grouped (deferred) weight gradients against one launch per layer, alternating
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
for b in 256 16; do
  for rep in 1 2; do
    for mode in grouped per_layer; do
      if [ $mode = per_layer ]; then export SRK_NO_DEFER_WGRAD=1; else unset SRK_NO_DEFER_WGRAD; fi
      python3 bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --sustain-seconds 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('batch %4d %-10s %9.1f p/s  %.3f ms/step  sustained %9.1f' % ($b, '$mode', d['value'], d['ms_per_step'], d['sustained_value']))"
    done
  done
done
