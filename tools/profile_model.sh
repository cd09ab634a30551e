#!/bin/bash
# ON THE GPU BOX: rocprofv3 --kernel-trace of `bench.py --model M --batch B` (no roofline / CPU legs), grouped per (kernel, grid):
# where a training step of that model goes.  usage: tools/profile_model.sh <model> <batch> [tag]  -> gpurun_out/<tag>_trace_<model>_b<batch>.txt
M=${1:-wdsr_b}; B=${2:-16}; TAG=${3:-r3}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_${M}_b$B; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --sustain-seconds 0 > "$OUT/trace.log" 2>&1
cd "$REPO"
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --sustain-seconds 0"
  echo "# (warm-up, capture-time eager steps and the 13 replays together; per (kernel, grid) group: launches, median, total)"
  tail -1 "$OUT/trace.log" | cut -c1-400
  python3 tools/kernel_groups.py "$OUT/trace" 28; } > gpurun_out/${TAG}_trace_${M}_b$B.txt
cat gpurun_out/${TAG}_trace_${M}_b$B.txt
rm -rf "$OUT/trace"
