#!/usr/bin/env python3
"""One training step of a bench.py kernel trace (the CSV tools/profile_step.sh keeps) in dispatch order: the launches between the last two
`adam_bump_kernel` dispatches.  usage: step_list.py <trace.csv.gz> [min_us]  (conv_ws launches shorter than min_us are folded into one line)"""
import csv, gzip, io, sys, collections
rows = list(csv.DictReader(io.TextIOWrapper(gzip.open(sys.argv[1]))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
minus = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
idx = [i for i, r in enumerate(rows) if "adam_bump" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
tot, fold = 0.0, collections.defaultdict(lambda: [0, 0.0])
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:64]
    if "conv_ws_kernel" in n and d < minus:
        fold[n][0] += 1; fold[n][1] += d
    else:
        print(f"{n:64s} {r['Grid_Size_X']:>9s} {d:9.1f}")
for n, (c, d) in fold.items():
    print(f"{n:64s} x{c:<8d} {d:9.1f}  ({d / c:.1f} each)")
print("sum of kernel durations %.1f us, span %.1f us" % (tot, (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3))
