#!/bin/bash
# A/B timing of the current library against tools/ubench/libsrk_prev.so (a build of an earlier commit), same box,
# alternating 3 x per configuration, best of each
for args in "--n 256 --relu 1 --res 0" "--n 256 --relu 0 --res 1" "--n 64 --relu 1 --res 0" "--n 64 --relu 0 --res 1" "--n 64 --hw 96 --cin 64 --cout 256 --relu 0"; do
  bp=999999; bn=999999
  for rep in 1 2 3; do
    for lib in prev new; do
      if [ $lib = prev ]; then export SRK_LIB_PATH=$PWD/tools/ubench/libsrk_prev.so; else unset SRK_LIB_PATH; fi
      us=$(python tools/microbench_conv.py $args --iters 40 2>&1 | grep -v amdgpu | sed -n 's/.*: \([0-9.]*\) us\/iter.*/\1/p')
      if [ $lib = prev ]; then bp=$(python3 -c "print(min($bp,$us))"); else bn=$(python3 -c "print(min($bn,$us))"); fi
    done
  done
  echo "$args : prev $bp us  new $bn us  ($(python3 -c "print(round(100*($bn/$bp-1),1))") %)"
done
