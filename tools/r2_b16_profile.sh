#!/bin/bash
# ON THE GPU BOX: where does a batch-16 training step go?  Kernel trace (durations + idle gaps) of the hipGraph
# step at batch 16.  usage: tools/r2_b16_profile.sh [models...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
for m in ${@:-edsr_baseline rcan}; do
  OUT=$REPO/gpurun_out/r2_b16_trace_$m; mkdir -p "$OUT"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --model $m --batch 16 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --sustain-seconds 0 > "$OUT/trace.log" 2>&1 )
  python3 tools/kernel_groups.py "$OUT/trace" 30 > "$OUT/groups.txt" 2>&1
  cat "$OUT/groups.txt"
  rm -rf "$OUT/trace"/*/*.db 2>/dev/null
done
