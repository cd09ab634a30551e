#!/bin/bash
# ON THE GPU BOX: bench lines of every model at a given batch.  usage: tools/r2_sweep.sh <batch> [models...]
B=${1:-16}; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
for m in ${@:-edsr_baseline rcan edsr_large wdsr_b rdn_b srresnet ddbpn}; do
  python3 bench.py --model $m --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-roofline --sustain-seconds 1 > gpurun_out/r2_sweep_${m}_$B.json 2> gpurun_out/r2_sweep_${m}_$B.err
  tail -1 gpurun_out/r2_sweep_${m}_$B.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-14s b%-4d %10.1f p/s  %9.3f ms/step  sustained %10.1f  model_mfma_frac %.3f  loss %.4f' % ('$m', $B, d['value'], d['ms_per_step'], d.get('sustained_value', 0), d['model_mfma_frac'], d['config']['loss_after_timed_steps']))" || tail -3 gpurun_out/r2_sweep_${m}_$B.err
done
