#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
python3 -m pytest tests/test_gpu_round2.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -60
