cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hr_tail.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python bench.py --batch 16 --steps 50 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
SRK_DEBUG=1 SRK_NO_SIDE_STREAM=1 timeout 600 python bench.py --batch 16 --steps 50 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python bench.py --batch 16 --steps 50 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
SRK_DEBUG=1 SRK_NO_SIDE_STREAM=1 timeout 600 python bench.py --batch 16 --steps 50 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python bench.py --model rcan --batch 16 --steps 30 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
