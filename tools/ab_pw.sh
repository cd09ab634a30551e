#!/bin/bash
# ON THE GPU BOX: same-box A/B of round 5's two WDSR-B changes at batch $1 (default 16):
#   old swizzle  = tools/ubench/libsrk_pwold.so (pw_chain.hip built with -DSRK_PW_OLD_SWZ=1: the h / gh images with the activation swizzle)
#   grouped finalize = SRK_DEBUG=1 SRK_PW_GROUP_FIN=1 (ONE finalize launch for all pointwise pairs of a backward pass; default: one per pair)
B=${1:-16}
run() { env "$@" python bench.py --model wdsr_b --batch $B --steps 40 --warmup 10 --no-cpu-baseline --no-other-configs --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for r in 1 2; do
  echo "round-4 form (old swizzle, finalize per pair):       $(run SRK_LIB_PATH=$PWD/tools/ubench/libsrk_pwold.so)"
  echo "round 5 default (new swizzle, finalize per pair):    $(run X=1)"
  echo "old swizzle, ONE grouped finalize:                   $(run SRK_LIB_PATH=$PWD/tools/ubench/libsrk_pwold.so SRK_DEBUG=1 SRK_PW_GROUP_FIN=1)"
  echo "new swizzle, ONE grouped finalize:                   $(run SRK_DEBUG=1 SRK_PW_GROUP_FIN=1)"
done
