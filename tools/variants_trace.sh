#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel-trace summaries of ISOLATED launches of every flavour of the dominant kernel at the
# bench shape (64->64 3x3 bf16 @48x48): conv+bias+ReLU, conv*scale+residual, data-gradient with ReLU mask, grouped weight
# gradient (8 layers per flush) + grouped finalize.  One rocprofv3 run per flavour, program directly behind `--`.
# usage: tools/variants_trace.sh [batch]   -> gpurun_out/variants_n<batch>.txt
N=${1:-256}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/variants_n$N; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in plain residual mask wgrad; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$v" -- python3 "$REPO/tools/microbench_variants.py" --n $N --variant $v --iters 30 > "$OUT/$v.log" 2>&1
done
python3 - "$OUT" "$N" <<'PY'
import csv, glob, os, sys
out, n = sys.argv[1], int(sys.argv[2])
fl = 2.0 * n * 48 * 48 * 64 * 64 * 9
with open(out + ".txt", "w") as fh:
    fh.write(f"isolated launches, 64->64 3x3 bf16 @48x48 x{n}: rocprofv3 --kernel-trace --stats per flavour (tools/variants_trace.sh {n})\n")
    fh.write(f"{'flavour':10s} {'kernel':72s} {'calls':>6s} {'avg_us':>9s} {'min_us':>8s} {'TFLOP/s':>9s} {'frac 2.5PF':>10s}\n")
    for v in ("plain", "residual", "mask", "wgrad"):
        rows = []
        for f in glob.glob(os.path.join(out, v, "**", "*kernel_stats.csv"), recursive=True):
            rows += list(csv.DictReader(open(f)))
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            name = r["Name"]
            if not any(k in name for k in ("conv_ws_kernel", "conv_wgrad_ws_group", "wgrad_finalize_group")):
                continue
            avg = float(r["AverageNs"]) / 1e3
            per_layer = avg / 8 if "group" in name else avg          # the grouped launches carry 8 layers
            tf = fl / per_layer / 1e6 if "finalize" not in name else 0.0
            fh.write(f"{v:10s} {name[:72]:72s} {r['Calls']:>6s} {avg:9.2f} {float(r['MinNs'])/1e3:8.2f} {tf:9.1f} {tf/2500:10.3f}\n")
print(open(out + ".txt").read())
PY
