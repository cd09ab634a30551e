python -m pytest tests/test_gpu_hr_tail.py tests/test_gpu_round3.py -x -q -m gpu 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-250
tools/profile_step.sh edsr_baseline 256 r4c
