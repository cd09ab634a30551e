#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
python3 -m pytest tests/test_gpu_wgrad_group.py tests/test_gpu_models.py -m gpu -x -q 2>&1 | tail -3
bash tools/r2_b16_profile.sh edsr_baseline 2>&1 | grep -E "finalize|wgrad_ws_group|multi_tensor" | cut -c1-170
python3 bench.py --batch 16 --steps 30 --warmup 5 --no-cpu-baseline --sustain-seconds 1 --no-roofline | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['config']['workload'], d['value'], d['ms_per_step'])"
