#!/bin/bash
# ON THE GPU BOX: evidence for the small-batch pair kernel: (1) rocprofv3 --kernel-trace --stats of tools/microbench_pair.py (chains of
# 32 blocks, pair vs the two launches it replaces, 16 / 32 / 64 patches), (2) s_memtime anatomy of workgroup 0 in the three modes
# (diagnostics build).  -> gpurun_out/r2p_pair_profile.txt
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pairprof; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/tools/microbench_pair.py" 16 32 64 > "$OUT/micro.log" 2>&1
cd "$REPO"
{
  echo "# tools/microbench_pair.py 16 32 64 under rocprofv3 --kernel-trace --stats (launch period per block of a 32-block chain, hipGraph replays)"
  grep "us per block" "$OUT/micro.log"
  echo
  echo "# rocprofv3 kernel stats of the same run (conv_pair_kernel vs conv_ws_kernel variants; averages over all three batch sizes)"
  python3 - "$OUT/trace" <<'PY'
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:6]:
    print(f"{r['Name'][:90]:90s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:8.2f} us  min {float(r['MinNs'])/1e3:8.2f} us")
PY
  echo
  echo "# s_memtime anatomy of workgroup 0 (make -C sr-pytorch-lightning_amd/csrc stamp; tools/stamp_pair.py 16), ticks ~ cycles"
  for m in 0 1 2; do echo "## STAMP_CA=$m (0 plain ResBlock forward, 1 RCAB backward with the CALayer backward on the way in + pooling, 2 RCAB forward with the previous block's CALayer on the way in + pooling)"; STAMP_CA=$m python3 tools/stamp_pair.py 16 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r2p_pair_profile.txt
cat gpurun_out/r2p_pair_profile.txt | cut -c1-220
rm -rf "$OUT/trace"
