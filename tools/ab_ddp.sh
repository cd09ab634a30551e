#!/bin/bash
# ON THE GPU BOX: what the multi-rank step structure costs before any communication: the default single-process line against a forced
# 1-rank process group (SRK_FORCE_DDP=1: GradSync buckets, graph segments, RCCL all-reduce of one rank), same box, batch 256 and 16.
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 256 16; do for v in 0 1 0 1; do
  SRK_FORCE_DDP=$v python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 1 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b SRK_FORCE_DDP=$v', d['value'], d['ms_per_step'], 'sustained', d.get('sustained_value'), d['config']['hip_graph'], d['config']['grad_sync'])"
done; done
