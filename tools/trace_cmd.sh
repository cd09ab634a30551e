#!/bin/bash
# ON THE GPU BOX: kernel-trace + stats of an arbitrary python script; prints per-kernel calls/avg.
# usage: tools/trace_cmd.sh <tag> script.py [args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/$1" "${@:2}" > "$OUT/trace.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
with open(os.path.join(out, "kernel_stats_summary.txt"), "w") as fh:
    fh.write(f"{'kernel':80s} {'calls':>7s} {'total_ms':>9s} {'avg_us':>9s} {'min_us':>8s} {'pct':>6s}\n")
    for r in rows[:30]:
        fh.write(f"{r['Name'][:80]:80s} {r['Calls']:>7s} {float(r['TotalDurationNs'])/1e6:9.3f} {float(r['AverageNs'])/1e3:9.2f} {float(r['MinNs'])/1e3:8.2f} {float(r['Percentage']):6.2f}\n")
print(open(os.path.join(out, "kernel_stats_summary.txt")).read())
PY
