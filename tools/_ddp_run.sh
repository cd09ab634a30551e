#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "gradsync or trainer" 2>&1 | tail -5
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "bench_runs or oracle_trajectory" 2>&1 | tail -3
tools/ab_ddp.sh
SRK_FORCE_DDP=1 python bench.py --batch 16 --steps 40 --warmup 10 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-1200
