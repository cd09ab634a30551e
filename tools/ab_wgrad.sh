#!/bin/bash
# A/B timing of the weight-gradient kernel: current library vs tools/ubench/libsrk_prev.so, same box, alternating, best of 3
for args in "--n 256" "--n 64" "--n 64 --hw 96 --cin 64 --cout 256"; do
  bp=999999; bn=999999
  for rep in 1 2 3; do
    for lib in prev new; do
      if [ $lib = prev ]; then export SRK_LIB_PATH=$PWD/tools/ubench/libsrk_prev.so; else unset SRK_LIB_PATH; fi
      us=$(python tools/microbench_conv.py --mode wgrad $args --iters 40 2>&1 | grep -v amdgpu | sed -n 's/.*: \([0-9.]*\) us\/iter.*/\1/p')
      if [ $lib = prev ]; then bp=$(python3 -c "print(min($bp,$us))"); else bn=$(python3 -c "print(min($bn,$us))"); fi
    done
  done
  echo "wgrad $args : prev $bp us  new $bn us  ($(python3 -c "print(round(100*($bn/$bp-1),1))") %)"
done
