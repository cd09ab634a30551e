#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "16 256 256" "16 128 64" "16 512 64" "64 256 256" "256 256 256"; do
  set -- $cfg
  for v in 0 1; do
    echo -n "n=$1 cin=$2 cout=$3 SRK_NO_KS=$v: "
    SRK_NO_KS=$v python3 tools/microbench_conv.py --n $1 --cin $2 --cout $3 2>/dev/null | tail -1
  done
done
bash tools/r2_b16_profile.sh edsr_large 2>&1 | head -14
