#!/usr/bin/env python3
"""One training step in dispatch order from a rocprofv3 --kernel-trace run of bench.py: the LAST `period` dispatches, where the period is
found as the shortest repeat of the kernel-name sequence at the end of the trace (graph replays issue the same launches every step).
usage: step_sequence.py <dir> [steps_to_average]   -> one line per launch: index, kernel, grid, LDS, mean us over the last steps"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
avg = int(sys.argv[2]) if len(sys.argv) > 2 else 5
names = [(r["Kernel_Name"], r.get("Grid_Size_X"), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", ""))) for r in rows]
n = len(names)
period = None
for p in range(8, n // (avg + 1)):
    if all(names[n - 1 - i] == names[n - 1 - i - p] for i in range(p * avg)):
        period = p
        break
if period is None:
    sys.exit("no repeating step found in %d dispatches" % n)
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = 0.0
span = []
for s in range(avg):
    blk = rows[n - (s + 1) * period:n - s * period]
    span.append((int(blk[-1]["End_Timestamp"]) - int(blk[0]["Start_Timestamp"])) / 1e3)
print("# %d launches per step, averaged over the last %d steps; step span (first start -> last end) %.1f us" % (period, avg, sum(span) / avg))
for i in range(period):
    us = sum(dur(rows[n - (s + 1) * period + i]) for s in range(avg)) / avg
    tot += us
    k = names[n - period + i]
    short = k[0].replace("(anonymous namespace)::", "").replace("void ", "")[:72]
    print("%4d %-72s grid %-8s lds %-7s %9.2f us" % (i, short, k[1], k[2], us))
print("# sum of kernel durations %.1f us" % tot)
