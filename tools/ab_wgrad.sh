#!/bin/bash
# A/B timing of the weight-gradient kernel: current library vs tools/ubench/libsrk_prev.so, same box
for args in "--n 256" "--n 64" "--n 64 --hw 96 --cin 64 --cout 256"; do
  for lib in prev new; do
    if [ $lib = prev ]; then export SRK_LIB_PATH=$PWD/tools/ubench/libsrk_prev.so; else unset SRK_LIB_PATH; fi
    echo -n "$lib: "; python tools/microbench_conv.py --mode wgrad $args --iters 40 2>&1 | grep -v amdgpu
  done
done
