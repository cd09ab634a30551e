cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$*', '->', d['value'], d['unit'], d['ms_per_step'], 'ms/step', 'mfma_frac', d.get('model_mfma_frac'))"; }
run --model rcan --batch 128 --steps 4 --warmup 2
run --model rcan --batch 256 --steps 3 --warmup 2
run --model rdn_b --batch 64 --steps 4 --warmup 2
run --model wdsr_b --batch 64 --steps 4 --warmup 2
run --model wdsr_b --batch 256 --steps 4 --warmup 2
