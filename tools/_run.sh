python -m pytest tests/test_gpu_round5.py tests/test_gpu_data_metrics.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_models.py -x -q -m gpu -k "trajectory or rdn_b" 2>&1 | tail -3
for r in 1 2; do for v in 0 1; do
SRK_DEBUG=1 SRK_NO_L1_FUSED_MEAN=$v python bench.py --batch 16 --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('two-launch L1=$v', d['value'], d['ms_per_step'])"
done; done
