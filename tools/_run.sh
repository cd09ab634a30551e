python -m pytest tests/test_gpu_conv_pair.py tests/test_gpu_round5.py -x -q -m gpu 2>&1 | tail -2
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "rcan_64_feature" 2>&1 | tail -2
bash tools/ab_lib.sh
for m in 2 1; do echo "== STAMP_CA=$m"; STAMP_CA=$m python tools/stamp_pair.py 16 2>&1 | grep -v amdgpu.ids | cut -c1-900; done
