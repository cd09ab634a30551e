#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "16 128 768" "16 768 112" "16 112 768" "16 768 128" "16 576 64" "16 64 64" "16 256 64"; do set -- $cfg; for v in 0 1; do echo -n "n=$1 cin=$2 cout=$3 SRK_NO_P1=$v: "; SRK_NO_P1=$v python3 tools/microbench_conv.py --n $1 --cin $2 --cout $3 --k 1 2>/dev/null | tail -1; done; done
