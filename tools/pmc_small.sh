#!/bin/bash
# ON THE GPU BOX: clock + MFMA-busy for the graph-replayed conv microbench.  usage: pmc_small.sh tag [microbench args]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tr" -- python3 "$REPO/tools/microbench_conv.py" "$@" > "$OUT/tr.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d "$OUT/pm" -- python3 "$REPO/tools/microbench_conv.py" "$@" > "$OUT/pm.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, os, collections
out=sys.argv[1]
dur={}
for f in glob.glob(os.path.join(out,"tr","**","*kernel_stats.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Name"]: dur[r["Name"][:50]]=float(r["AverageNs"])
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out,"pm","**","*counter_collection.csv"),recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]: agg[r["Kernel_Name"][:50]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    m={c:sum(v)/len(v) for c,v in d.items()}
    ns=dur.get(k,0)
    print(k, "avg_us=%.2f"%(ns/1e3))
    if ns:
        print("   eff clock GHz = %.3f" % (m.get("GRBM_GUI_ACTIVE",0)/8/ns))
        print("   MFMA busy frac (per SIMD, at that clock) = %.3f" % (m.get("SQ_VALU_MFMA_BUSY_CYCLES",0)/1024/(m.get("GRBM_GUI_ACTIVE",1)/8)))
    for c,v in sorted(m.items()): print("   %-28s %.0f"%(c,v))
PY
