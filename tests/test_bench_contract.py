"""bench.py's contract with the driver (no GPU needed): flags, the JSON line's keys / types (checked on the committed
line of the final tree, profiles/r1_final_bench_default.json), and the refusal to run without an MI355X (no CPU
fallback that would print a number)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_flags_exist():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--batch", "--model", "--dtype", "--inference"):
        assert flag in out.stdout


def test_committed_line_has_the_contract_keys():
    line = json.loads(open(os.path.join(ROOT, "profiles", "r1_final_bench_default.json")).readline())
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                 ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(line[k], t), k
    assert line["vs_baseline"] is None and line["scaling"] == "weak" and line["data"] == "synthetic"
    assert "workload" in line["config"] and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    c = line["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    # value is whole-job throughput: global batch * steps / time
    assert abs(line["value"] - line["config"]["global_batch"] / (line["ms_per_step"] * 1e-3)) / line["value"] < 1e-3


def test_refuses_to_run_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert not out.stdout.strip().startswith("{"), "no JSON line (no number) without the GPU"
