#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_models.py tests/test_gpu_wgrad_group.py -m gpu -x -q 2>&1 | tail -5
for cfg in "rcan 16" "edsr_baseline 16"; do
  set -- $cfg
  python3 bench.py --model $1 --batch $2 --steps 30 --warmup 5 --no-cpu-baseline --sustain-seconds 1 --no-roofline > gpurun_out/r2_chk_$1_$2.json 2> gpurun_out/r2_chk_$1_$2.err
  tail -1 gpurun_out/r2_chk_$1_$2.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d.get('sustained_value'), d['config']['loss_after_timed_steps'])" || tail -5 gpurun_out/r2_chk_$1_$2.err
done
bash tools/r2_b16_profile.sh rcan 2>&1 | head -16 | cut -c1-160
