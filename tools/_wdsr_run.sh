python -m pytest tests/test_gpu_pw_chain.py -x -q -m gpu 2>&1 | tail -3
for b in 16 256; do python bench.py --model wdsr_b --batch $b --steps 20 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200; done
tools/profile_step.sh wdsr_b 256 r4c
tools/profile_step.sh wdsr_b 16 r4c
