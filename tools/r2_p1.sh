#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_conv_ks.py tests/test_gpu_ops.py tests/test_gpu_models.py -x -q 2>&1 | tail -6
bash tools/ab_env.sh SRK_NO_P1 wdsr_b 16 2
bash tools/ab_env.sh SRK_NO_P1 rdn_b 16 1
