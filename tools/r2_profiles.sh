#!/bin/bash
# ON THE GPU BOX: everything profiles/ quotes for round 2, from the tree as it is (run last; copy gpurun_out/r2p_* into profiles/).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
O=gpurun_out
python3 bench.py > $O/r2p_bench_default.json 2> $O/r2p_bench_default.err; tail -1 $O/r2p_bench_default.json | cut -c1-300
python3 bench.py --batch 16 --steps 50 --warmup 10 > $O/r2p_bench_b16_edsr_baseline.json 2>/dev/null; tail -1 $O/r2p_bench_b16_edsr_baseline.json | cut -c1-200
python3 bench.py --batch 16 --steps 50 --warmup 10 --model rcan > $O/r2p_bench_b16_rcan.json 2>/dev/null; tail -1 $O/r2p_bench_b16_rcan.json | cut -c1-200
bash tools/profile_bench.sh r2final > /dev/null 2>&1; cp $O/prof_r2final/kernel_stats_summary.txt $O/r2p_kernel_stats_default.txt; head -12 $O/r2p_kernel_stats_default.txt | cut -c1-160
bash tools/variants_trace.sh 256 > /dev/null 2>&1; cp $O/variants_n256.txt $O/r2p_variants_n256.txt; cat $O/r2p_variants_n256.txt | cut -c1-170
bash tools/variants_trace.sh 16 > /dev/null 2>&1; cp $O/variants_n16.txt $O/r2p_variants_n16.txt; cat $O/r2p_variants_n16.txt | cut -c1-170
bash tools/pmc_traffic.sh 256 16 > /dev/null 2>&1; cp $O/r2_pmc_traffic.json $O/r2p_pmc_traffic.json; cat $O/r2p_pmc_traffic.json | head -30
for m in edsr_baseline rcan; do bash tools/r2_b16_profile.sh $m > $O/r2p_b16_trace_$m.txt 2>&1; head -14 $O/r2p_b16_trace_$m.txt | cut -c1-160; done
rm -rf $O/prof_r2final/trace $O/variants_n256 $O/variants_n16 $O/pmc_traffic 2>/dev/null
