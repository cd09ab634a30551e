#!/bin/bash
# A/B timing of the current library against tools/ubench/libsrk_prev.so (a build of an earlier commit), same box
for args in "--n 256 --relu 1 --res 0" "--n 256 --relu 0 --res 1" "--n 64 --relu 1 --res 0" "--n 64 --relu 0 --res 1" "--n 64 --hw 96 --cin 64 --cout 256 --relu 0"; do
  for lib in prev new; do
    if [ $lib = prev ]; then export SRK_LIB_PATH=$PWD/tools/ubench/libsrk_prev.so; else unset SRK_LIB_PATH; fi
    echo -n "$lib: "; python tools/microbench_conv.py $args --iters 40 2>&1 | grep -v amdgpu
  done
done
