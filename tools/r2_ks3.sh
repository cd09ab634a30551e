#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_conv_ks.py tests/test_gpu_ops.py tests/test_gpu_models.py -x -q 2>&1 | tail -4
bash tools/ab_env.sh SRK_NO_KS wdsr_b 16 2
