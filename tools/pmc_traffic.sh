#!/bin/bash
# ON THE GPU BOX: HBM-side traffic per launch of the kernels bench.py quotes, from rocprofv3 PMC passes -- FETCH_SIZE and WRITE_SIZE in
# SEPARATE runs, no trace domains, the program directly behind `--` (MI355X_MICROARCH.md "HBM" / "rocprofv3 PMC slots") -- written to
# gpurun_out/<tag>_pmc_traffic.json (tag = $SRK_PROFILE_TAG, default r6) with the fingerprint of each kernel's generated ISA
# (csrc/kernel_isa.json: bench.py only quotes a figure whose fingerprint matches the instructions it is timing).
#   1. the isolated conv + bias + ReLU launch of conv_ws_kernel (tools/microbench_variants.py)          key conv_bias_relu:edsr_baseline:64x48x<n>xbf16
#   2. the dominant kernel of every benched configuration INSIDE its training step (bench.py under the profiler)   key step:<model>:b<batch>:bf16
# usage: tools/pmc_traffic.sh [batch sizes of part 1...]   (default: 256)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
export SRK_PROFILE_TAG=${SRK_PROFILE_TAG:-r6}
OUT=$REPO/gpurun_out/pmc_traffic; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for n in ${@:-256}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/n${n}_$c" -- python3 "$REPO/tools/microbench_variants.py" --n $n --variant plain --iters 5 > "$OUT/n${n}_$c.log" 2>&1
  done
done
for mb in "edsr_baseline 256" "edsr_baseline 16" "rcan 16" "edsr_large 16" "wdsr_b 16" "rdn_b 16" "srresnet 16" "ddbpn 16"; do set -- $mb
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/step_$1_b$2_$c" -- python3 "$REPO/bench.py" --model $1 --batch $2 --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > "$OUT/step_$1_b$2_$c.log" 2>&1
  done
done
python3 - "$OUT" "$REPO" <<'PY'
import csv, glob, json, os, sys
out, repo = sys.argv[1], sys.argv[2]
sys.path.insert(0, repo)
import bench


def mean_counter(d, kernel, c):
    v = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == c:
                v.append(float(r["Counter_Value"]))
    return (sum(v) / len(v) if v else None), len(v)


tab = {}
for d in sorted(glob.glob(os.path.join(out, "n*_FETCH_SIZE"))):
    n = int(os.path.basename(d).split("_")[0][1:])
    fe, k = mean_counter(d, "conv_ws_kernel", "FETCH_SIZE")
    wr, _ = mean_counter(os.path.join(out, f"n{n}_WRITE_SIZE"), "conv_ws_kernel", "WRITE_SIZE")
    if fe is None or wr is None:
        continue
    tab[f"conv_bias_relu:edsr_baseline:64x48x{n}xbf16"] = {
        "fetch_kib": fe, "write_kib": wr, "launches": k, "isa_key": "conv_ws_plain_bf16", "isa_sha": bench.kernel_fingerprint("conv_ws_plain_bf16"),
        "source": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) -- python3 tools/microbench_variants.py --n %d --variant plain --iters 5 (conv + bias + ReLU + sign bits: the launch a per-layer training step issues)" % n,
        "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "algorithmic_bytes_per_launch": 2.0 * n * 48 * 48 * 64 * 2 + n * 48 * 48 * 8}
# the dominant kernel of each configuration in its step: (kernel-name substring, fingerprint key, algorithmic bytes per launch, what they are)
P = 48 * 48
T64 = lambda n: n * P * 64 * 2.0          # one 64-channel 16-bit activation tensor
DOM = {
    ("edsr_baseline", 256): ("conv_trunk_kernel", "conv_trunk_bf16", None, "33 layers per launch: see roofline.in_step.hbm_view (input, output, residual where the flavour has one, sign bits)"),
    ("edsr_baseline", 16): ("conv_pair_kernel", "conv_pair_bf16", 3 * T64(16), "x read, the intermediate and the output written (training keeps the intermediate)"),
    ("rcan", 16): ("conv_pair_kernel", "conv_pair_bf16", 5 * T64(16), "RCAB pair with the neighbouring block's channel attention: t and x of the previous block read, the block input, the intermediate and conv 2's output written"),
    ("srresnet", 16): ("conv_ws_kernel<0, 2, 4, true, false, 0>", "conv_ws_plain_bf16", 2 * T64(16), "input read, output written (a BatchNorm sits between the convs of a block: single launches)"),
    ("edsr_large", 16): ("conv_ks_kernel<0, false>", "conv_ks_bf16", 2 * 4 * T64(16) + 9 * 256 * 256 * 2, "256-channel input read, output written, weights"),
    ("rdn_b", 16): ("conv_ks_kernel<0, false>", "conv_ks_bf16", (4.5 + 1) * T64(16), "RDN-B dense layers: Cin = 64 + 64 c, c = 0..7 (288 on average) read, 64 channels written"),
    ("wdsr_b", 16): ("pw_fwd_kernel", "pw_fwd_bf16", 16 * P * (128 + 112) * 2.0, "128-channel block input read, 112 (102 used) channels written; the 768-channel tensor never leaves the chip"),
    ("ddbpn", 16): ("proj_up_kernel", "proj_up_bf16", 16 * (P * 16 + P) * 32 * 2.0, "32-channel LR tensor read, 32-channel HR tensor written"),
}
for (m, b), (kn, key, alg, what) in DOM.items():
    fe, k = mean_counter(os.path.join(out, f"step_{m}_b{b}_FETCH_SIZE"), kn, "FETCH_SIZE")
    wr, _ = mean_counter(os.path.join(out, f"step_{m}_b{b}_WRITE_SIZE"), kn, "WRITE_SIZE")
    if fe is None or wr is None:
        print("no counters for", m, b, kn, file=sys.stderr)
        continue
    tab[f"step:{m}:b{b}:bf16"] = {
        "kernel": kn, "fetch_kib": fe, "write_kib": wr, "launches": k, "isa_key": key, "isa_sha": bench.kernel_fingerprint(key),
        "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "algorithmic_bytes_per_launch": alg, "algorithmic_bytes_are": what,
        "source": f"tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate runs) -- python3 bench.py --model {m} --batch {b} --steps 3 --warmup 2 "
                  "--no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0; mean over every dispatch of the kernel in the run (eager warm-up steps and graph replays)"}
json.dump(tab, open(os.path.join(repo, "gpurun_out", os.environ.get("SRK_PROFILE_TAG", "r6") + "_pmc_traffic.json"), "w"), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ("kernel", "hbm_bytes_per_launch", "algorithmic_bytes_per_launch", "launches") if kk in v} for k, v in tab.items()}, indent=1))
PY
