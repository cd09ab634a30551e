#!/bin/bash
# ON THE GPU BOX: one bench line per model at the given batch sizes (training step as one hipGraph, no roofline / CPU legs).
# usage: tools/sweep.sh "<batches>" [models...]   e.g. tools/sweep.sh "16 64 256" edsr_baseline rcan   -> gpurun_out/sweep.txt
B=${1:-16}; shift
MODELS=${@:-edsr_baseline rcan edsr_large wdsr_b rdn_b srresnet ddbpn}
cd ${GRAFT_REPO_ROOT:-/root/repo}
for m in $MODELS; do for b in $B; do
  python bench.py --model $m --batch $b --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 1 2>/dev/null | tail -1 |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(f\"{sys.argv[1]:14s} b{sys.argv[2]:>4s}: {d['value']:9.1f} patches/s {d['ms_per_step']:8.3f} ms/step  sustained {d.get('sustained_value', 0):9.1f}  model_mfma_frac {d['model_mfma_frac']:.4f}\")" $m $b
done; done | tee gpurun_out/sweep.txt
