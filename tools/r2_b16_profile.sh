#!/bin/bash
# ON THE GPU BOX: where does a batch-16 training step go?  Kernel trace (durations + idle gaps) of the hipGraph
# step for EDSR-baseline and RCAN at batch 16, plus the un-profiled bench lines.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
for m in edsr_baseline rcan; do
  python3 bench.py --model $m --batch 16 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/r2_b16_$m.json 2> gpurun_out/r2_b16_$m.err
  tail -1 gpurun_out/r2_b16_$m.json | cut -c1-400
  OUT=$REPO/gpurun_out/r2_b16_trace_$m; mkdir -p "$OUT"
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --model $m --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > "$OUT/trace.log" 2>&1 )
  python3 tools/kernel_groups.py "$OUT/trace" 24 > "$OUT/groups.txt" 2>&1
  python3 tools/gap_stats.py "$OUT/trace" 1500 > "$OUT/gaps.txt" 2>&1
  cat "$OUT/groups.txt" "$OUT/gaps.txt"
  rm -rf "$OUT/trace"/*/*.db 2>/dev/null
done
