python -m pytest tests/test_gpu_hr_tail.py -x -q -m gpu 2>&1 | tail -3
python bench.py --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
python bench.py --batch 16 --steps 50 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
tools/profile_step.sh edsr_baseline 16 r4c
tools/profile_step.sh edsr_baseline 256 r4c
