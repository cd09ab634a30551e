#!/usr/bin/env python3
"""Static instruction mix of one kernel in a hipcc -save-temps .s file, split at barriers / MFMA region.
usage: isa_stats.py file.s kernel_substring"""
import collections, re, sys
S, key = sys.argv[1], sys.argv[2]
lines = open(S).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0]][0]
end = [i for i, l in enumerate(lines) if i > start and ".amdhsa_kernel" in l][0]
body = lines[start:end]
bar = [i for i, l in enumerate(body) if "s_barrier" in l]
mf = [i for i, l in enumerate(body) if "v_mfma" in l]
print("kernel lines", len(body), "barriers at", bar, "mfma first/last", mf[0], mf[-1])


def stats(a, b, name):
    c = collections.Counter()
    for l in body[a:b]:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.split(";")[0].strip().endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"): k = "mfma"
        elif op.startswith("v_accvgpr"): k = "accvgpr_mov"
        elif op.startswith("v_"): k = "v:" + re.sub(r"_e(32|64)$", "", op)[:22]
        elif op.startswith("s_"): k = "s:" + ("waitcnt" if "waitcnt" in op else ("branch" if "branch" in op else "other"))
        elif op.startswith("ds_"): k = "ds"
        else: k = op[:24]
        c[k] += 1
    nv = sum(v for k, v in c.items() if k.startswith("v:") or k == "accvgpr_mov")
    print(f"--- {name}: lines {a}-{b}  VALU={nv} total={sum(c.values())}")
    for k, v in c.most_common(14):
        print(f"   {k:30s} {v}")


cuts = [0] + bar + [mf[0], mf[-1] + 1, len(body)]
cuts = sorted(set(cuts))
for a, b in zip(cuts[:-1], cuts[1:]):
    stats(a, b, "region")
