#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_conv_pair.py tests/test_gpu_ops.py tests/test_gpu_models.py -x -q 2>&1 | tail -3
bash tools/ab_env.sh SRK_CA_UNFUSED rcan 16 2
STAMP_CA=1 python3 tools/stamp_pair.py 16 2>&1 | grep -v amdgpu.ids | head -3
STAMP_CA=2 python3 tools/stamp_pair.py 16 2>&1 | grep -v amdgpu.ids | head -3
