#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
python3 -m pytest tests/test_gpu_round2.py -m gpu -q -x -k "unfold or batchnorm or srresnet or ddbpn or psnr" -s 2>&1 | grep -v "^$" | tail -25
python3 -m pytest tests/test_gpu_models.py -m gpu -q -x -k "srresnet or ddbpn" 2>&1 | tail -12
