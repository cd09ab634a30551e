for r in 1 2; do
for lib in tools/ubench/libsrk_prev.so sr-pytorch-lightning_amd/libsrk_gfx950.so; do
SRK_LIB_PATH=$PWD/$lib python bench.py --model rcan --batch 16 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['roofline'].get('isolated', d['roofline']).get('variants_us'))"
done; done
