python -m pytest tests/test_gpu_conv_pair.py tests/test_gpu_ws_epilogue.py -x -q -m gpu 2>&1 | tail -3
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "rcan_64_feature" 2>&1 | tail -2
for r in 1 2; do for lib in tools/ubench/libsrk_prev.so sr-pytorch-lightning_amd/libsrk_gfx950.so; do echo $lib; SRK_LIB_PATH=$PWD/$lib python tools/microbench_pair.py 16 2>&1 | grep pair; done; done
for m in 0 2 1; do echo "== STAMP_CA=$m"; STAMP_CA=$m python tools/stamp_pair.py 16 2>&1 | grep -v amdgpu.ids | cut -c1-900; done
bash tools/ab_lib.sh
