#!/bin/bash
# ON THE GPU BOX: everything profiles/ carries for a round, from ONE box (so that bench.py's event timings and the rocprofv3
# averages can be compared): bench lines, rocprofv3 kernel traces, PMC passes (one counter group per run, FETCH_SIZE and
# WRITE_SIZE in separate passes, program directly behind `--`).   usage: tools/collect_profiles.sh [tag]   -> gpurun_out/<tag>_*
# Afterwards, here: copy gpurun_out/<tag>_* into profiles/ and run tools/gen_results.py (it rewrites the number tables of
# profiles/README.md and DESIGN.md from the files; tests/test_docs_numbers.py checks that they are in sync).
TAG=${1:-r6}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; O=gpurun_out
python __graft_entry__.py smoke > $O/${TAG}_smoke.txt 2>&1; tail -1 $O/${TAG}_smoke.txt
python -m pytest tests -q -m gpu 2>&1 | tail -2 > $O/${TAG}_gpu_tests.txt; cat $O/${TAG}_gpu_tests.txt
SRK_PROFILE_TAG=$TAG tools/pmc_traffic.sh 256 > /dev/null 2>&1; cp $O/${TAG}_pmc_traffic.json profiles/${TAG}_pmc_traffic.json      # (the bench line below quotes it: every configuration's dominant kernel in its step)
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 400 $O/${TAG}_bench_default.json; echo
python bench.py --dtype f16 --no-cpu-baseline --no-other-configs > $O/${TAG}_bench_default_f16.json 2>/dev/null
python bench.py --inference --no-cpu-baseline --no-other-configs --no-roofline > $O/${TAG}_bench_inference.json 2>/dev/null
tools/variants_trace.sh 256 $TAG > /dev/null 2>&1; cat $O/${TAG}_variants_n256.txt
for m in edsr_baseline rcan edsr_large wdsr_b rdn_b ddbpn srresnet; do
  python bench.py --model $m --batch 16 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/${TAG}_bench_b16_$m.json
  python3 -c "import json,sys; d=json.load(open('$O/${TAG}_bench_b16_$m.json')); print('$m b16', d['value'], d['roofline'].get('frac'), d['roofline'].get('isolated', d['roofline']).get('variants_us'))"
done
tools/profile_bench.sh ${TAG}final --no-roofline --sustain-seconds 0 > /dev/null 2>&1; cp $O/prof_${TAG}final/kernel_stats_summary.txt $O/${TAG}_kernel_stats_default.txt; head -14 $O/${TAG}_kernel_stats_default.txt | cut -c1-150
# one step in dispatch order (body / upsampler / HR stage / weight gradients are separable although the persistent kernels share a grid size)
for mb in "edsr_baseline 256" "edsr_baseline 16" "rcan 16" "rcan 64" "rcan 256" "wdsr_b 16" "wdsr_b 256"; do set -- $mb; tools/profile_step.sh $1 $2 $TAG > /dev/null 2>&1
  python3 tools/step_list.py $O/${TAG}_step_$1_b$2.csv.gz 25 > $O/${TAG}_step_$1_b$2.txt; rm -f $O/${TAG}_step_$1_b$2.csv.gz; done
SRK_DEBUG=1 SRK_NO_HR_COLLAPSE=1 tools/profile_step.sh edsr_baseline 256 ${TAG}lw > /dev/null 2>&1; python3 tools/step_list.py $O/${TAG}lw_step_edsr_baseline_b256.csv.gz 60 > $O/${TAG}_step_edsr_baseline_b256_layerwise.txt; rm -f $O/${TAG}lw_step_*
# the same default step with the trunk as one launch per convolution (SRK_DEBUG=1 SRK_NO_TRUNK=1): what the image-stationary launch replaced
SRK_DEBUG=1 SRK_NO_TRUNK=1 tools/profile_step.sh edsr_baseline 256 ${TAG}pl > /dev/null 2>&1; python3 tools/step_list.py $O/${TAG}pl_step_edsr_baseline_b256.csv.gz 25 > $O/${TAG}_step_edsr_baseline_b256_per_layer.txt; rm -f $O/${TAG}pl_step_*
{ echo "# tools/microbench_trunk.py: a 32-convolution residual chain at 256 x 48 x 48 bf16, per-layer srk_conv2d launches against ONE srk_conv_trunk launch (same box, hipGraph replays, HIP events)"
  python tools/microbench_trunk.py 2>&1 | grep -v amdgpu.ids
  echo "# SRK_NO_TRUNK=1 against the default, whole step (bench.py, two runs each)"
  for r in 1 2; do for v in 1 0; do echo "SRK_NO_TRUNK=$v $(SRK_DEBUG=1 SRK_NO_TRUNK=$v python bench.py --no-cpu-baseline --no-roofline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])")"; done; done; } > $O/${TAG}_ab_trunk.txt 2>&1; cat $O/${TAG}_ab_trunk.txt
tools/ab_exact_relu.sh > $O/${TAG}_exact_relu.txt 2>&1; cat $O/${TAG}_exact_relu.txt
for m in edsr_large rdn_b srresnet ddbpn; do tools/profile_model.sh $m 16 $TAG > /dev/null 2>&1; done
# PMC: the body kernels, the new 5x5 kernels, the kernels the verdict named
tools/pmc_kernel.sh ${TAG}_conv_trunk_n256 conv_trunk_kernel tools/microbench_trunk.py --seconds 0.05 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_conv_ws_plain_n256 conv_ws_kernel tools/microbench_variants.py --n 256 --variant plain --iters 5 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_conv_ws_residual_n256 conv_ws_kernel tools/microbench_variants.py --n 256 --variant residual --iters 5 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_wgrad_group_n256 conv_wgrad_ws_group_kernel tools/microbench_variants.py --n 256 --variant wgrad --iters 4 > /dev/null 2>&1
# (the 5x5 kernels and the mask variant did not change since round 4: profiles/r4_lk5_*_pmc.txt, r4_conv_ws_mask_n256_pmc.txt)
tools/pmc_kernel.sh ${TAG}_conv_ks_256_n16 conv_ks_kernel tools/microbench_conv.py --cin 256 --cout 256 --n 16 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_conv_pair_n16 conv_pair_kernel tools/microbench_pair.py 16 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_pw_wgrad_n256 pw_wgrad_kernel tools/microbench_pw.py --n 256 --only wgrad --iters 5 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_pw_wgrad_n16 pw_wgrad_kernel tools/microbench_pw.py --n 16 --only wgrad --iters 5 > /dev/null 2>&1
for f in $O/${TAG}_*_pmc.txt; do echo "== $f"; grep -E "^void|MFMA pipe|HBM-side|BANK_CONFLICT" $f | cut -c1-140; done
tools/ab_pw.sh 16 > $O/${TAG}_ab_pw_b16.txt 2>&1; cat $O/${TAG}_ab_pw_b16.txt
# in-kernel s_memtime anatomy (diagnostics build tools/ubench/libsrk_stamp.so = `make -C sr-pytorch-lightning_amd/csrc stamp`, built here, travels with the snapshot)
if [ -f tools/ubench/libsrk_stamp.so ]; then
  { for m in 0 2 1; do echo "== conv_pair_kernel, 16 x 48 x 48, STAMP_CA=$m (0: ResBlock; 2 / 1: RCAB forward / backward with the channel attention of the neighbouring block)"; STAMP_CA=$m python tools/stamp_pair.py 16 2>/dev/null | grep -v amdgpu.ids | cut -c1-900; done
    echo "== conv_ws_kernel, 256 x 48 x 48, conv + ReLU"; python tools/stamp_ws.py 256 2>/dev/null | cut -c1-300
    echo "== conv_ws_kernel, 256 x 48 x 48, * scale + residual (prefetch variant)"; STAMP_RES=1 python tools/stamp_ws.py 256 2>/dev/null | cut -c1-300
  } > $O/${TAG}_stamps.txt
fi
tools/ab_ddp.sh > $O/${TAG}_ab_ddp.txt 2>&1; cat $O/${TAG}_ab_ddp.txt
tools/sweep.sh "16 64 256" > /dev/null 2>&1; cp $O/sweep.txt $O/${TAG}_sweep.txt; cat $O/${TAG}_sweep.txt
