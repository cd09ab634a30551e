#!/bin/bash
# ON THE GPU BOX: rocprofv3 kernel-trace of ISOLATED, SUSTAINED launches of every flavour of the dominant kernel at the bench shape
# (64->64 3x3 bf16 @48x48): conv+bias+ReLU, conv*scale+residual, data-gradient with ReLU mask, grouped weight gradient (8 layers
# per flush) + grouped finalize.  One rocprofv3 run per flavour, program directly behind `--`; every run replays its launches
# for 1.5 s and the summary averages the SECOND HALF of the dispatches (by start time): the state bench.py's `variants_us` is
# quoted on (the clock drops within ~1 s under these kernels).
# usage: tools/variants_trace.sh [batch] [tag]   -> gpurun_out/<tag>_variants_n<batch>.txt
N=${1:-256}; TAG=${2:-r3}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/variants_n$N; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in plain residual mask wgrad; do
  rocprofv3 --kernel-trace --output-format csv -d "$OUT/$v" -- python3 "$REPO/tools/microbench_variants.py" --n $N --variant $v --iters 30 --seconds 1.5 > "$OUT/$v.log" 2>&1
done
python3 - "$OUT" "$N" "$REPO/gpurun_out/${TAG}_variants_n$N.txt" <<'PY'
import csv, glob, os, sys, collections
out, n, dst = sys.argv[1], int(sys.argv[2]), sys.argv[3]
fl = 2.0 * n * 48 * 48 * 64 * 64 * 9
with open(dst, "w") as fh:
    fh.write(f"isolated SUSTAINED launches (1.5 s of replays per flavour, second half of the dispatches averaged), 64->64 3x3 bf16 @48x48 x{n}:\n")
    fh.write(f"rocprofv3 --kernel-trace -- python3 tools/microbench_variants.py --n {n} --variant <v> --iters 30 --seconds 1.5   (tools/variants_trace.sh {n})\n")
    fh.write(f"{'flavour':10s} {'kernel':72s} {'calls':>7s} {'avg_us':>9s} {'first_half':>10s} {'min_us':>8s} {'TFLOP/s':>9s} {'frac 2.5PF':>10s}\n")
    for v in ("plain", "residual", "mask", "wgrad"):
        d = collections.defaultdict(list)
        for f in glob.glob(os.path.join(out, v, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                d[r["Kernel_Name"]].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        for name, rows in sorted(d.items(), key=lambda kv: -sum(x[1] for x in kv[1])):
            if not any(k in name for k in ("conv_ws_kernel", "conv_wgrad_ws_group", "wgrad_finalize_group")):
                continue
            rows.sort()
            h2 = [x[1] for x in rows[len(rows) // 2:]]
            h1 = [x[1] for x in rows[:len(rows) // 2]] or h2
            avg = sum(h2) / len(h2)
            per_layer = avg / 8 if "group" in name else avg          # the grouped launches carry 8 layers
            tf = fl / per_layer / 1e6 if "finalize" not in name else 0.0
            fh.write(f"{v:10s} {name[:72]:72s} {len(rows):7d} {avg:9.2f} {sum(h1)/len(h1):10.2f} {min(h2):8.2f} {tf:9.1f} {tf/2500:10.3f}\n")
print(open(dst).read())
PY
rm -rf "$OUT"
