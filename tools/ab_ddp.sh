#!/bin/bash
# ON THE GPU BOX: what the multi-rank step structure costs before any communication: the default single-process line against a forced
# 1-rank process group (SRK_FORCE_DDP=1: GradSync buckets, graph segments, RCCL all-reduce of one rank), same box, batch 256 and 16 -- and
# the TWO-SEGMENT backward (SRK_DDP_SEGMENTS=2: the later layers' weight gradients are flushed and their buckets' all-reduce is started
# before the earlier layers' backward runs, ops.backward_segments / trainer.OverlappedGraphStep), which is what overlaps EDSR-baseline's 6 MB of
# gradients with the rest of the backward pass on N > 1 GPUs (default for models above 48 MB only: trainer.auto_segments).
cd ${GRAFT_REPO_ROOT:-/root/repo}
for b in 256 16; do for v in "0 auto" "1 auto" "1 2" "0 auto" "1 auto" "1 2"; do set -- $v
  SRK_FORCE_DDP=$1 SRK_DDP_SEGMENTS=$2 python bench.py --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 1 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('batch $b SRK_FORCE_DDP=$1 SRK_DDP_SEGMENTS=$2', d['value'], d['ms_per_step'], 'sustained', d.get('sustained_value'), d['config']['hip_graph'], d['config']['grad_sync'], 'allreduce_ms', d['config'].get('allreduce_ms'))"
done; done
