#!/bin/bash
# ON THE GPU BOX: new tests first, then the full GPU suite, then the bench lines
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
python3 -m pytest tests/test_gpu_wgrad_group.py -x -q 2>&1 | tail -15
python3 -m pytest tests -m gpu -x -q 2>&1 | tail -15
for cfg in "edsr_baseline 16" "edsr_baseline 256" "rcan 16"; do
  set -- $cfg
  python3 bench.py --model $1 --batch $2 --steps 30 --warmup 5 --no-cpu-baseline --sustain-seconds 1 > gpurun_out/r2_chk_$1_$2.json 2> gpurun_out/r2_chk_$1_$2.err
  tail -1 gpurun_out/r2_chk_$1_$2.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], d['value'], d['ms_per_step'], d.get('sustained_value'), d.get('roofline',{}).get('variants_us'), d.get('roofline',{}).get('step_weighted_frac'), d.get('roofline',{}).get('frac'))" || tail -5 gpurun_out/r2_chk_$1_$2.err
done
