#!/bin/bash
# ON THE GPU BOX: the numbers of DESIGN.md section 7 (batch sweep, inference, the other BASELINE configs)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', '->', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'graph' if d['config'].get('hip_graph') else 'eager')"; }
for b in 16 64 128 256 384 512; do run --batch $b --steps 10 --warmup 3; done
run --inference --steps 10 --warmup 3
for m in rcan edsr_large wdsr_b rdn_b; do run --model $m --batch 16 --steps 5 --warmup 2; done
run --model rcan --batch 64 --steps 5 --warmup 2
run --model edsr_large --batch 64 --steps 5 --warmup 2
run --dtype f16 --steps 10 --warmup 3
