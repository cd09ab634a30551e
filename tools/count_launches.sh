cd /tmp && export TMPDIR=/tmp
for st in 10 50; do
rm -rf /tmp/tr; rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python3 $GRAFT_REPO_ROOT/bench.py --model rcan --batch 16 --steps $st --warmup 3 --no-cpu-baseline --no-roofline --no-other-configs --sustain-seconds 0 > /tmp/tr.log 2>&1
python3 - <<PY
import csv, glob, collections
d = collections.Counter()
for f in glob.glob("/tmp/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"][:50]] += 1
print("steps $st:", {k: v for k, v in d.items() if "copyBuffer" in k or "upload" in k or "pack_kernel" in k or "conv_pair" in k})
PY
done
