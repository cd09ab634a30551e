cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_hr_tail.py tests/test_gpu_round4.py -x -q -m gpu 2>&1 | tail -2
mkdir -p gpurun_out/hrk
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hrk -o n16 -- python3 tools/microbench_hrtail.py --n 16 --iters 20 > gpurun_out/hrk/n16.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/hrk -o n256 -- python3 tools/microbench_hrtail.py --n 256 --iters 10 > gpurun_out/hrk/n256.log 2>&1
for f in n16 n256; do echo "== $f"; python3 - <<PY
import csv,glob
fn=glob.glob("gpurun_out/hrk/**/${f}_kernel_stats.csv", recursive=True)
rows=list(csv.DictReader(open(fn[0])))
for r in rows[:14]:
    if "lk5" in r["Name"] or "lk_conv" in r["Name"] or "finalize" in r["Name"]: print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"])
PY
done
timeout 600 python bench.py --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
timeout 600 python bench.py --batch 16 --steps 50 --no-cpu-baseline --no-other-configs --no-roofline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-200
