#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): kernel-trace + stats of the default bench, then HBM PMC passes for the
# dominant kernel in separate runs (never --pmc together with --stats / trace domains).
# usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-r1}; shift || true
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$REPO/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs "$@" > "$OUT/trace.log" 2>&1
tail -2 "$OUT/trace.log"
# summarise: per-kernel calls / total / average
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
files = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
with open(os.path.join(out, "kernel_stats_summary.txt"), "w") as fh:
    fh.write(f"{'kernel':90s} {'calls':>8s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>7s}\n")
    for r in rows[:40]:
        fh.write(f"{r['Name'][:90]:90s} {r['Calls']:>8s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:10.2f} {float(r['Percentage']):7.2f}\n")
print(open(os.path.join(out, "kernel_stats_summary.txt")).read())
PY
