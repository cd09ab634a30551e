#!/usr/bin/env python3
"""Per (kernel, grid, LDS) group durations from a rocprofv3 --kernel-trace run.  usage: kernel_groups.py <dir> [top]"""
import csv, glob, os, sys, collections, statistics
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
d = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:64], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    d[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(k, len(v), "median %.1f us total %.2f ms" % (statistics.median(v), sum(v) / 1e3))
