#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_ops.py tests/test_gpu_conv_ks.py tests/test_gpu_conv_pair.py tests/test_gpu_models.py -x -q 2>&1 | tail -4
for m in edsr_large edsr_baseline; do
python3 bench.py --model $m --batch 16 --steps 30 --warmup 5 --no-roofline --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$m b16', d['value'], d['ms_per_step'])"
done
bash tools/r2_b16_profile.sh edsr_large 2>&1 | grep -i "pack\|conv_ks\|wgrad" | head
