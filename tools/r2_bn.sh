#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_round2.py tests/test_gpu_models.py -x -q 2>&1 | tail -4
bash tools/r2_sweep.sh 16 srresnet ddbpn 2>&1 | tail -2
