#!/bin/bash
# ON THE GPU BOX: HBM traffic of the dominant kernel (conv_ws_kernel, 64->64 3x3 bf16 @48x48, conv+bias+ReLU) per launch
# from rocprofv3 PMC passes -- FETCH_SIZE and WRITE_SIZE in SEPARATE runs, no trace domains (MI355X_MICROARCH.md "HBM" /
# "rocprofv3 PMC slots") -- written to gpurun_out/<tag>_pmc_traffic.json (tag = $SRK_PROFILE_TAG, default r5) together with the fingerprint of the kernel's
# generated ISA (csrc/kernel_isa.json; bench.py only quotes a figure whose fingerprint matches the instructions it is timing).
# usage: tools/pmc_traffic.sh [batch sizes...]   (default: 256)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export SRK_PROFILE_TAG=${SRK_PROFILE_TAG:-r5}
OUT=$REPO/gpurun_out/pmc_traffic; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in ${@:-256}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/n${n}_$c" -- python3 "$REPO/tools/microbench_variants.py" --n $n --variant plain --iters 5 > "$OUT/n${n}_$c.log" 2>&1
  done
done
python3 - "$OUT" "$REPO" <<'PY'
import csv, glob, json, os, sys
out, repo = sys.argv[1], sys.argv[2]
sys.path.insert(0, repo)
import bench
tab = {}
for d in sorted(glob.glob(os.path.join(out, "n*_FETCH_SIZE"))):
    n = int(os.path.basename(d).split("_")[0][1:])
    vals = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v = []
        for f in glob.glob(os.path.join(out, f"n{n}_{c}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "conv_ws_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    v.append(float(r["Counter_Value"]))
        vals[c] = sum(v) / max(len(v), 1)
        vals[c + "_n"] = len(v)
    tab[f"conv_bias_relu:edsr_baseline:64x48x{n}xbf16"] = {"fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"], "launches": vals["FETCH_SIZE_n"],
                             "isa_key": "conv_ws_plain_bf16", "isa_sha": bench.kernel_fingerprint("conv_ws_plain_bf16"),
                             "source": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) -- python3 tools/microbench_variants.py --n %d --variant plain --iters 5 (conv + bias + ReLU + sign bits: the launch a training step issues)" % n,
                             "hbm_bytes_per_launch": (2 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024,
                             "algorithmic_bytes_per_launch": 2.0 * n * 48 * 48 * 64 * 2 + n * 48 * 48 * 8}
json.dump(tab, open(os.path.join(repo, "gpurun_out", os.environ.get("SRK_PROFILE_TAG", "r5") + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps(tab, indent=1))
PY
