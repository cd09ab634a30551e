#!/bin/bash
# ON THE GPU BOX: cost of the NaN-preserving ReLU build (SRK_EXACT_RELU=1 -> libsrk_gfx950_exact.so, csrc/srk_common.h) against the default build,
# same box: the default line and the batch-16 lines of the models whose epilogues carry a ReLU.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "edsr_baseline 256" "edsr_baseline 16" "rcan 16" "wdsr_b 16"; do set -- $cfg
  for r in 1 2; do for v in 0 1; do
    SRK_EXACT_RELU=$v python bench.py --model $1 --batch $2 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 |
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 b$2 SRK_EXACT_RELU=$v', d['value'], d['ms_per_step'])"
  done; done
done
