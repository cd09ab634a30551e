#!/bin/bash
# ON THE GPU BOX: PMC counters of the launches of ONE kernel, one counter group per rocprofv3 run (--pmc only: no trace domains,
# FETCH_SIZE and WRITE_SIZE in separate passes: MI355X_MICROARCH.md "rocprofv3 PMC slots"), program directly after `--`.
# usage: tools/pmc_kernel.sh <tag> <kernel name substring> <script.py> [script args...]  -> gpurun_out/<tag>_pmc.txt
TAG=$1; KN=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 "$REPO/$1" "${@:2}" > "$OUT/g$i.log" 2>&1
done
cd "$REPO"
python3 - "$OUT" "$KN" "$TAG" "$*" <<'PY' > gpurun_out/${TAG}_pmc.txt
import csv, glob, sys, os, collections, hashlib
out, kn, tag, cmd = sys.argv[1:5]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(out, "g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if kn in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
h = hashlib.sha256()
for f in sorted(glob.glob("sr-pytorch-lightning_amd/csrc/*.hip")) + ["sr-pytorch-lightning_amd/csrc/srk_common.h"]:
    h.update(open(f, "rb").read())
print(f"# rocprofv3 --pmc <group> -- python3 {cmd}   (one run per counter group; kernels matching '{kn}'; csrc fingerprint {h.hexdigest()[:16]})")
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):16.1f}")
    g = lambda n: sum(d[n]) / len(d[n]) if n in d else None
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
        # MFMA_BUSY counts cycles per SIMD summed over 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
        print(f"   -> MFMA pipe busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (g('GRBM_GUI_ACTIVE') / 8):.3f} of the launch's cycles")
    if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
        print(f"   -> HBM-side traffic per launch: {(2 * g('FETCH_SIZE') + g('WRITE_SIZE')) * 1024 / 1e6:.1f} MB (FETCH_SIZE x 2 + WRITE_SIZE, KiB)")
PY
cat gpurun_out/${TAG}_pmc.txt
rm -rf "$OUT"
