#!/bin/bash
# ON THE GPU BOX: everything profiles/ carries for a round, from ONE box (so that bench.py's event timings and the rocprofv3
# averages can be compared): bench lines, rocprofv3 kernel traces, PMC passes (one counter group per run, FETCH_SIZE and
# WRITE_SIZE in separate passes, program directly behind `--`).   usage: tools/collect_profiles.sh [tag]   -> gpurun_out/<tag>_*
TAG=${1:-r4}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; O=gpurun_out
python bench.py > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err; tail -c 600 $O/${TAG}_bench_default.json; echo
tools/variants_trace.sh 256 $TAG > /dev/null 2>&1; cat $O/${TAG}_variants_n256.txt
tools/pmc_traffic.sh 256 > /dev/null 2>&1; cat $O/r4_pmc_traffic.json | head -12
for m in edsr_baseline rcan edsr_large wdsr_b rdn_b ddbpn srresnet; do
  python bench.py --model $m --batch 16 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 > $O/${TAG}_bench_b16_$m.json
  python3 -c "import json,sys; d=json.load(open('$O/${TAG}_bench_b16_$m.json')); print('$m b16', d['value'], d['roofline']['variants_us'], d['roofline'].get('step_weighted_frac'))"
done
tools/profile_bench.sh ${TAG}final --no-roofline --sustain-seconds 0 > /dev/null 2>&1; cp $O/prof_${TAG}final/kernel_stats_summary.txt $O/${TAG}_kernel_stats_default.txt; head -12 $O/${TAG}_kernel_stats_default.txt | cut -c1-150
for m in edsr_baseline rcan edsr_large wdsr_b rdn_b srresnet ddbpn; do tools/profile_model.sh $m 16 $TAG > /dev/null 2>&1; done
tools/pmc_kernel.sh ${TAG}_conv_pair_n16 conv_pair_kernel tools/microbench_pair.py 16 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_conv_ks_256_n16 conv_ks_kernel tools/microbench_conv.py --n 16 --cin 256 --cout 256 --iters 5 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_conv1x1_576_n16 conv1x1_kernel tools/microbench_conv.py --n 16 --cin 576 --cout 64 --k 1 --iters 5 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_wgrad_group_n256 conv_wgrad_ws_group_kernel tools/microbench_variants.py --n 256 --variant wgrad --iters 4 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_conv_ws_n256 conv_ws_kernel tools/microbench_variants.py --n 256 --variant residual --iters 5 > /dev/null 2>&1
for k in fwd2 bwd wgrad; do tools/pmc_kernel.sh ${TAG}_pw_${k%2}_n256 pw_${k}_kernel tools/microbench_pw.py --n 256 --only ${k%2} --iters 5 > /dev/null 2>&1; done
for n in 16 64 256; do python tools/microbench_proj.py --n $n --iters $((n > 64 ? 5 : 20)); done > $O/${TAG}_proj_microbench.txt 2>/dev/null; python tools/microbench_proj.py --n 16 --prelu >> $O/${TAG}_proj_microbench.txt 2>/dev/null; cat $O/${TAG}_proj_microbench.txt
tools/pmc_kernel.sh ${TAG}_lk_conv_rows lk_conv_rows_kernel bench.py --model srresnet --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /dev/null 2>&1
tools/pmc_kernel.sh ${TAG}_lk_wgrad_allrows lk_wgrad_allrows_kernel bench.py --model srresnet --batch 16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /dev/null 2>&1
for k in up down wgrad; do tools/pmc_kernel.sh ${TAG}_proj_${k}_n16 proj_${k}_kernel tools/microbench_proj.py --n 16 --only $k --iters 5 > /dev/null 2>&1; done
for f in $O/${TAG}_*_pmc.txt; do echo "== $f"; grep -E "^void|MFMA pipe|HBM-side|BANK_CONFLICT" $f | cut -c1-140; done
tools/ab_ddp.sh > $O/${TAG}_ab_ddp.txt 2>&1; cat $O/${TAG}_ab_ddp.txt
tools/sweep.sh "16 64 256" > /dev/null 2>&1; cp $O/sweep.txt $O/${TAG}_sweep.txt; cat $O/${TAG}_sweep.txt
