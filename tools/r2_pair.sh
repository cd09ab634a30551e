#!/bin/bash
# conv_pair: parity tests, microbench, batch-16 benches with and without the pair
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pair
timeout 600 python3 -m pytest tests/test_gpu_conv_pair.py -x -q 2>&1 | tail -15
timeout 300 python3 tools/microbench_pair.py 16 32 64 2>&1 | tail -8
for m in edsr_baseline rcan; do
  for off in 0 1; do
    SRK_NO_PAIR=$off timeout 600 python3 bench.py --model $m --batch 16 --steps 30 --warmup 5 --no-roofline > gpurun_out/pair/${m}_b16_off$off.json 2>gpurun_out/pair/${m}_b16_off$off.err
    python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/pair/${m}_b16_off$off.json").read().strip().splitlines()[-1])
    print("$m off=$off", d["value"], d["ms_per_step"])
except Exception as e:
    print("$m off=$off failed", e); print(open("gpurun_out/pair/${m}_b16_off$off.err").read()[-1500:])
PY
  done
done
