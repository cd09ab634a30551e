#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_conv_ks.py tests/test_gpu_ops.py -x -q 2>&1 | tail -8
bash tools/ab_env.sh SRK_NO_KS edsr_large 16 2
bash tools/ab_env.sh SRK_NO_KS rdn_b 16 1
