python -m pytest tests/test_gpu_round2.py tests/test_gpu_round4.py tests/test_gpu_wgrad_group.py -x -q -m gpu 2>&1 | tail -3
for m in "rcan 16" "edsr_baseline 16" "wdsr_b 16" "edsr_baseline 256"; do set -- $m; for r in 1 2; do for v in 1 0; do
SRK_DEBUG=1 SRK_NO_STAGED_TABLES=$v python bench.py --model $1 --batch $2 --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 b$2 kernel-argument uploads=$v', d['value'], d['ms_per_step'])"
done; done; done
