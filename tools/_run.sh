python -m pytest tests/test_gpu_conv_pair.py -x -q -m gpu 2>&1 | tail -1
for r in 1 2; do for lib in tools/ubench/libsrk_prev.so tools/ubench/libsrk_headA.so sr-pytorch-lightning_amd/libsrk_gfx950.so tools/ubench/libsrk_st1.so tools/ubench/libsrk_st2.so; do
SRK_LIB_PATH=$PWD/$lib python bench.py --model rcan --batch 16 --steps 60 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])"
done; done
