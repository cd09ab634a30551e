#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace run.
usage: gap_stats.py <dir with *kernel_trace.csv> [n_last_kernels]
Prints, over the last N dispatches (default: all): busy time, idle time between dispatches, gap histogram and
the kernels that follow the largest gaps."""
import csv, glob, os, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    with open(f) as fh:
        rows += list(csv.DictReader(fh))
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows)
rows = rows[-n:]
busy = sum(e - s for s, e, _ in rows)
span = rows[-1][1] - rows[0][0]
gaps = [(rows[i + 1][0] - rows[i][1], rows[i][2][:50], rows[i + 1][2][:50]) for i in range(len(rows) - 1)]
idle = sum(max(g, 0) for g, _, _ in gaps)
print(f"{len(rows)} dispatches, span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle between dispatches {idle/1e6:.3f} ms ({100*idle/span:.1f}%)")
h = collections.Counter()
for g, _, _ in gaps:
    k = "<0" if g < 0 else "0-1us" if g < 1000 else "1-2us" if g < 2000 else "2-4us" if g < 4000 else "4-8us" if g < 8000 else "8-50us" if g < 50000 else ">50us"
    h[k] += 1
print("gap histogram:", dict(h))
byk = collections.defaultdict(lambda: [0, 0])
for g, a, b in gaps:
    if g < 50000:
        byk[b][0] += 1; byk[b][1] += max(g, 0)
print("idle before kernel (top 12 by total):")
for k, (c, t) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"  {k:52s} n={c:5d} total {t/1e3:9.1f} us  avg {t/c/1e3:6.2f} us")
