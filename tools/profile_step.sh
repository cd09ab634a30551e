#!/bin/bash
# ON THE GPU BOX: rocprofv3 --kernel-trace of `bench.py --model M --batch B`; prints ONE step in dispatch order (tools/step_sequence.py) so
# the stages of the step (body, HR stage, weight gradients) can be told apart although the persistent kernels share one grid size.
# usage: tools/profile_step.sh <model> <batch> [tag]  -> gpurun_out/<tag>_step_<model>_b<batch>.txt
M=${1:-edsr_baseline}; B=${2:-256}; TAG=${3:-r4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_${TAG}_step_${M}_b$B; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > "$OUT/trace.log" 2>&1
cd "$REPO"
{ echo "# rocprofv3 --kernel-trace -- python3 bench.py --model $M --batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0"
  tail -1 "$OUT/trace.log" | cut -c1-300
  python3 tools/step_sequence.py "$OUT/trace" 5 2>&1; } > gpurun_out/${TAG}_step_${M}_b$B.txt
cat $(find "$OUT/trace" -name "*kernel_trace.csv" | head -1) | gzip -9 > gpurun_out/${TAG}_step_${M}_b$B.csv.gz
rm -rf "$OUT/trace"
