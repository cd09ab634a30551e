#!/usr/bin/env python3
"""Benchmark of the SR hot path on MI355X.

Workload (BASELINE.json configs[1]): EDSR-baseline (16 resblocks, 64 feats, res_scale 0.1) x4, synthetic
48x48 LR patches with 192x192 HR targets, bf16 storage / fp32 accumulate, one full TRAINING step per
"step": forward + L1 loss + backward + Adam update (the reference's training_step/configure_optimizers,
models/srmodel.py:145-171).  `value` = LR patches/s over all ranks with inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (the dominant kernel, the 64->64 3x3 implicit-GEMM
conv, timed with HIP events on its own stream inside this process) and, at N=1, `cpu_baseline` (the CPU
oracle's training step timed on this host's cores over a bounded sample).
"""
import argparse
import contextlib
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODELS = {
    # name: (class, ctor kwargs, conv GFLOP per patch forward, feats)
    "edsr_baseline": ("EDSR", dict(n_feats=64, n_resblocks=16, res_scale=0.1), 9.139, 64),
    "edsr_large": ("EDSR", dict(n_feats=256, n_resblocks=32, res_scale=0.1), 231.564, 256),
    "rcan": ("RCAN", dict(n_feats=64, reduction=16, n_resgroups=10, n_resblocks=20), 73.350, 64),
    "wdsr_b": ("WDSR", dict(type="B"), 21.974, 128),
    "rdn_b": ("RDN", dict(rdn_config="B"), 104.737, 64),
}
PEAK_TFLOPS = {"bf16": 2500.0, "f16": 2500.0, "f32": 157.3}     # dense MFMA, MI355X_MICROARCH.md
PREC = {"bf16": "bf16", "f16": 16, "f32": 32}
TDT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--batch", type=int, default=256, help="LR patches per GPU per step (weak scaling)")
    p.add_argument("--model", default="edsr_baseline", choices=sorted(MODELS))
    p.add_argument("--dtype", default="bf16", choices=sorted(PREC))
    p.add_argument("--patch", type=int, default=48, help="LR patch edge")
    p.add_argument("--scale", type=int, default=4)
    p.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--inference", action="store_true", help="forward only (patches/s of SRModel.forward)")
    return p.parse_args()


def cpu_baseline(model_name, scale, patch, seconds=15.0, max_steps=8):
    """The CPU oracle's training step (fp32, torch CPU, all host cores) on a bounded sample."""
    from oracle import train as OT
    cls, kw, _, _ = MODELS[model_name]
    # torch's CPU conv scales to ~32 threads on the GPU node's host and collapses beyond (measured: 4x64x48x48
    # conv 0.20 ms at 32 threads, 4.0 ms at 128), so the baseline uses min(32, available cores)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(32, avail))
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    m = OT.OracleModel(cls, scale_factor=scale, **kw)
    opt = m.configure_optimizers()[0]
    n = 16
    g = torch.Generator().manual_seed(1234)
    batch = {"lr": torch.rand(n, 3, patch, patch, generator=g), "hr": torch.rand(n, 3, patch * scale, patch * scale, generator=g)}

    def step():
        opt.zero_grad()
        m.training_step(batch)["loss"].backward()
        opt.step()
    step()                                   # warm-up
    t0 = time.perf_counter()
    k = 0
    while k < max_steps and (k == 0 or time.perf_counter() - t0 < seconds):
        step()
        k += 1
    dt = time.perf_counter() - t0
    return {"value": round(n * k / dt, 3), "unit": "LR patches/s", "cores": cores, "kind": "port",
            "sample": f"{k} training steps of batch {n} ({cls} fp32, torch {torch.__version__} CPU, {cores} threads), {dt:.1f} s"}


def dominant_kernel_roofline(A, batch, patch, feats, dtype, iters=100):
    """Average duration of ONE launch of the dominant kernel (F->F 3x3 conv + bias + ReLU on [batch, patch, patch, F]).
    `iters` launches are captured into one hipGraph and replayed between two HIP events on the launch stream, so
    the Python launch rate (~10 us) does not enter; achieved = algorithmic FLOPs per launch / duration."""
    dt = TDT[dtype]
    dev = torch.device("cuda", torch.cuda.current_device())
    x = (torch.rand(batch, patch, patch, feats, device=dev) - 0.5).to(dt)
    w = torch.nn.Parameter((torch.rand(feats, feats, 3, 3, device=dev) - 0.5) * 0.05)
    b = torch.nn.Parameter(torch.zeros(feats, device=dev))
    pk = A.ops.pack_conv(w, b, dt)
    out = torch.empty_like(x)
    kw = dict(N=batch, H=patch, W=patch, Cin=feats, Cout=feats, out=out, relu=True)
    for _ in range(5):
        A.ops.conv_raw(x, pk, **kw)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        A.ops.conv_raw(x, pk, **kw)
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            A.ops.conv_raw(x, pk, **kw)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    flops = 2.0 * batch * patch * patch * feats * feats * 9
    ach = flops / (us * 1e-6) / 1e12
    peak = PEAK_TFLOPS[dtype]
    esz = 4 if dtype == "f32" else 2
    alg_bytes = 2.0 * batch * patch * patch * feats * esz            # one read + one write of the activation
    return {"bound": "mfma", "kernel": f"conv_ws_kernel 3x3 {feats}->{feats} @{patch}x{patch} x{batch} ({dtype}), fwd = dgrad kernel",
            "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
            "us_per_launch": round(us, 2), "flops_per_launch": flops,
            "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps": round(alg_bytes / (us * 1e-6) / 1e9, 1),
            "traffic": TRAFFIC_PMC.get((feats, patch, batch, dtype))}


# HBM bytes per launch of the dominant kernel from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, KiB -> B;
# MI355X_MICROARCH.md "HBM"), measured with tools/pmc_conv.sh and committed under profiles/; None where not measured
TRAFFIC_PMC = {
    (64, 48, 64, "bf16"): (10656.0 * 2 + 18432.0) * 1024,      # profiles/r1_pmc_n64_summary.txt
    (64, 48, 256, "bf16"): (39550.3 * 2 + 73728.0) * 1024,     # profiles/r1_final_pmc_conv_n256.txt
}


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X; no GPU visible", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import sr_amd as A
    from sr_amd import trainer as T
    A._lib.load()
    force_ddp = os.environ.get("SRK_FORCE_DDP") == "1"      # 1-rank process group: exercises the DDP path on one GPU
    if world > 1 or force_ddp:
        T.init_distributed("cuda", force=force_ddp)

    cls, kw, gflop_fwd, feats = MODELS[a.model]
    torch.manual_seed(0)                                  # identical weights on every rank
    model = getattr(A, cls)(scale_factor=a.scale, precision=PREC[a.dtype], **kw).to(dev)
    batch = T.synthetic_batch(a.batch, 3, a.patch, a.scale, 1234 + rank, dev)
    params = [p for p in model.parameters() if p.requires_grad]
    ddp = world > 1 or force_ddp
    # hipGraph: one graph per training step.  With DDP the documented recipe applies (torch "CUDA graphs" notes): build
    # DDP and run >= 11 warm-up iterations on the side stream that also captures, so that RCCL's all-reduce and the
    # reducer's AccumulateGrad hooks are bound to it.  That works on a 1-rank RCCL group
    # (SRK_FORCE_DDP=1 SRK_BENCH_GRAPH_DDP=1: 21.5k patches/s vs 21.7k eager -- at batch 256 the GPU is the bottleneck either way), but an N > 1 capture cannot be tried on the 1-GPU
    # development box, so N > 1 launches eagerly unless SRK_BENCH_GRAPH_DDP=1.
    want_graph = not a.no_graph and (not ddp or os.environ.get("SRK_BENCH_GRAPH_DDP") == "1")
    side = torch.cuda.Stream() if want_graph else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        # gradient averaging: trainer.GradSync (one flat all-reduce after backward) unless SRK_USE_TORCH_DDP=1
        use_torch_ddp = os.environ.get("SRK_USE_TORCH_DDP") == "1"
        net = T.wrap_ddp(model, dev, force=force_ddp) if use_torch_ddp else model
        gsync = None
        if ddp and not use_torch_ddp:
            gsync = T.GradSync(model)
            gsync.broadcast()
        # reference optimizer: Adam at torch defaults (srmodel.py:145-154,602-603).  Same update rule, torch's fused
        # multi-tensor implementation (one kernel for all 74 tensors instead of ~150 tiny per-tensor launches),
        # capturable so the step counter lives on the device for hipGraph replay
        opt = torch.optim.Adam(params, fused=True, capturable=want_graph)

    def train_step():
        opt.zero_grad(set_to_none=True)
        sr = net(batch["lr"])
        loss = model._calculate_losses(img_sr=sr, img_hr=batch["hr"])["loss"]
        loss.backward()
        if gsync is not None:
            gsync.sync()
        opt.step()
        return loss

    def infer_step():
        with torch.no_grad():
            return net(batch["lr"])

    step = infer_step if a.inference else train_step
    last = {}
    graph = None
    used_graph = False
    if want_graph:
        try:
            with torch.cuda.stream(side):
                for _ in range(11 if ddp else 3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            # thread_local: RCCL's watchdog thread polls events while this thread captures
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local" if ddp else "global"):
                last["out"] = step()
            used_graph = True
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graph = None
            torch.cuda.synchronize()

    def run_one():
        if graph is not None:
            graph.replay()
        else:
            last["out"] = step()

    for _ in range(a.warmup):
        run_one()
    torch.cuda.synchronize()
    loss_first = float(last["out"].detach().float().mean()) if not a.inference and "out" in last else None
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        run_one()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())

    loss_last = float(last["out"].detach().float().mean()) if not a.inference and "out" in last else None
    if rank == 0:
        total = a.batch * world * a.steps
        value = total / el
        flop_per_patch = gflop_fwd * (1.0 if a.inference else 3.0) * (a.patch / 48.0) ** 2
        out = {
            "metric": "SR training throughput, LR patches/s (48x48 LR x4, fwd+bwd+Adam)" if not a.inference
                      else "SR inference throughput, LR patches/s (48x48 LR x4, forward)",
            "value": round(value, 2), "unit": "LR patches/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(el / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"{a.model} x{a.scale}, {a.patch}x{a.patch} LR patches, batch {a.batch}/GPU, "
                                   f"{'forward only' if a.inference else 'train step (L1 + Adam)'}",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "hip_graph": used_graph,
                       "grad_sync": (None if not ddp else "torch_ddp" if use_torch_ddp else "flat_allreduce"),
                       "loss_after_warmup": loss_first, "loss_after_timed_steps": loss_last},
            "model_mfma_frac": round(value / world * flop_per_patch / 1e3 / PEAK_TFLOPS[a.dtype], 4),
        }
        try:
            out["roofline"] = dominant_kernel_roofline(A, a.batch, a.patch, feats, a.dtype)
        except Exception as e:  # noqa: BLE001
            out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a.model, a.scale, a.patch)
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"}
        # RCCL prints a version banner through C stdio (buffered until exit when stdout is a pipe): flush it first so
        # that the JSON line is the last line of stdout
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
