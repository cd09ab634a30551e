#!/bin/bash
# best-of-3 alternating A/B between two library files: ab3.sh <libA> <libB> <microbench args...>
A=$1; B=$2; shift 2
ba=999999; bb=999999
for rep in 1 2 3; do
  for lib in A B; do
    if [ $lib = A ]; then export SRK_LIB_PATH=$A; else export SRK_LIB_PATH=$B; fi
    us=$(python tools/microbench_conv.py "$@" --iters 60 2>&1 | grep -v amdgpu | sed -n 's/.*: \([0-9.]*\) us\/iter.*/\1/p')
    if [ $lib = A ]; then ba=$(python3 -c "print(min($ba,$us))"); else bb=$(python3 -c "print(min($bb,$us))"); fi
  done
done
echo "$* : A $ba us  B $bb us  ($(python3 -c "print(round(100*($bb/$ba-1),1))") %)"
