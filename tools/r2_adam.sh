#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/adam
timeout 900 python3 -m pytest tests/test_gpu_optim.py tests/test_gpu_conv_pair.py tests/test_gpu_ops.py tests/test_gpu_models.py -x -q 2>&1 | tail -15
for m in ${MODELS:-rcan edsr_baseline}; do
  for b in 16; do
      timeout 900 python3 bench.py --model $m --batch $b --steps 30 --warmup 5 --no-roofline --no-cpu-baseline > gpurun_out/adam/${m}_b${b}.json 2>gpurun_out/adam/${m}_b${b}.err
      python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/adam/${m}_b${b}.json").read().strip().splitlines()[-1])
    print("$m b=$b", d["value"], d["ms_per_step"])
except Exception as e:
    print("$m b=$b failed", e); print(open("gpurun_out/adam/${m}_b${b}.err").read()[-1500:])
PY
  done
done
