for r in 1 2; do
for lib in sr-pytorch-lightning_amd/libsrk_gfx950.so tools/ubench/libsrk_noslp.so; do
SRK_LIB_PATH=$PWD/$lib python bench.py --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$lib b256', d['value'], r['frac'], r['in_step']['fwd_us_per_conv'], r['in_step']['dgrad_us_per_conv'], r['in_step']['wgrad_us_per_layer'], r.get('isolated',{}).get('variants_us'))"
done; done
for m in rcan wdsr_b edsr_baseline; do for lib in sr-pytorch-lightning_amd/libsrk_gfx950.so tools/ubench/libsrk_noslp.so; do
SRK_LIB_PATH=$PWD/$lib python bench.py --model $m --batch 16 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $m b16', d['value'])"
done; done
