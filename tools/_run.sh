python -m pytest tests/test_gpu_conv_pair.py -x -q -m gpu 2>&1 | tail -2
SRK_LIB_PATH=$PWD/tools/ubench/libsrk_pd3.so python -m pytest tests/test_gpu_conv_pair.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2 3; do for lib in sr-pytorch-lightning_amd/libsrk_gfx950.so tools/ubench/libsrk_pd3.so; do echo $lib; SRK_LIB_PATH=$PWD/$lib python tools/microbench_pair.py 16 2>&1 | grep "pair  "; done; done
for lib in sr-pytorch-lightning_amd/libsrk_gfx950.so tools/ubench/libsrk_pd3.so; do
SRK_LIB_PATH=$PWD/$lib python bench.py --model rcan --batch 16 --steps 50 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['value'])"
done
