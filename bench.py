#!/usr/bin/env python3
"""Benchmark of the SR hot path on MI355X.

Workload (BASELINE.json configs[1]): EDSR-baseline (16 resblocks, 64 feats, res_scale 0.1) x4, synthetic
48x48 LR patches with 192x192 HR targets, bf16 storage / fp32 accumulate, one full TRAINING step per
"step": forward + L1 loss + backward + Adam update (the reference's training_step/configure_optimizers,
models/srmodel.py:145-171).  `value` = LR patches/s over all ranks with inputs resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N ...          (WORLD_SIZE unset: starts N ranks itself through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  Extra objects: `roofline` (the dominant kernel, the 64->64 3x3 implicit-GEMM
conv, timed with HIP events on its own stream inside this process) and, at N=1, `cpu_baseline` (the CPU
oracle's training step timed on this host's cores over a bounded sample).
"""
import argparse
import contextlib
import hashlib
import json
import os
import socket
import subprocess
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
    # SURVEY.md 8(f) rank 4 (conv FLOPs counted on the reference's module graph, ConvTranspose2d included)
    "srresnet": ("SRResNet", dict(), 10.221, 64),
    "ddbpn": ("DDBPN", dict(), 11.506, 64),
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
    p.add_argument("--sustain-seconds", type=float, default=2.0,
                   help="after the timed steps, keep stepping this long and report sustained_ms_per_step (0 = skip)")
    p.add_argument("--no-roofline", action="store_true")
    return p.parse_args()


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks (one per GPU) as CHILD processes through
    torch.distributed.run and exit with their code.  Runs before this process has touched the GPU
    (torch.cuda.device_count() does not initialise it); nothing is ever exec'd over an initialised process."""
    vis = torch.cuda.device_count()
    if vis < a.gpus:
        print(f"bench.py --gpus {a.gpus}: only {vis} GPU(s) visible", file=sys.stderr)
        sys.exit(2)
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.exit(subprocess.call(cmd, env=env))


def cpu_baseline(model_name, scale, patch, seconds=15.0, max_steps=8, model=None, lr=None, hr=None):
    """The CPU oracle's training step (fp32, torch CPU, host cores) on a bounded sample -- and, since this is the one
    leg of the bench that may touch the oracle, the parity of the build against it on the bench's own weights and inputs:
    the oracle forward (fp32) of the first patches with the HIP model's CURRENT parameters, against the HIP forward."""
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
    out = {"value": round(n * k / dt, 3), "unit": "LR patches/s", "cores": cores, "kind": "port",
           "sample": f"{k} training steps of batch {n} ({cls} fp32, torch {torch.__version__} CPU, {cores} threads), {dt:.1f} s"}
    if model is not None and lr is not None:
        try:
            import math
            np_ = min(2, lr.shape[0])
            with torch.no_grad():
                y_hip = model(lr[:np_]).float().cpu()
                m.load_state_dict({k_: v.detach().float().cpu() for k_, v in model.state_dict().items()})
                y_ref = m.forward(lr[:np_].float().cpu())
            mse = float(((y_hip - y_ref) ** 2).mean())
            tgt = hr[:np_].float().cpu().clamp(0, 1)

            def ps(y):
                return 10.0 * math.log10(1.0 / max(float(((y.clamp(0, 1) - tgt) ** 2).mean()), 1e-12))
            out["parity"] = {"patches": np_, "psnr_build_vs_oracle_db": round(10.0 * math.log10(1.0 / max(mse, 1e-20)), 2),
                             "max_abs_err": float((y_hip - y_ref).abs().max()),
                             "delta_psnr_db": round(ps(y_hip) - ps(y_ref), 5),
                             "note": "HIP forward in the bench dtype vs fp32 CPU oracle, the model's current weights, synthetic (uniform) patches"}
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def _time_replays(fn, iters):
    """Average duration (us) of one call of `fn`: `iters` calls are captured into one hipGraph and replayed between two
    HIP events on the launch stream, so the Python launch rate (~10 us) does not enter."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def csrc_sha():
    """Fingerprint of the kernel sources a PMC measurement belongs to (profiles/r2_pmc_traffic.json carries the same)."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sr-pytorch-lightning_amd", "csrc")
    for f in ("conv_igemm.hip", "srk_common.h"):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def pmc_traffic(feats, patch, batch, dtype):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 +
    WRITE_SIZE, KiB -> B; MI355X_MICROARCH.md "HBM"), written by tools/pmc_traffic.sh.  None unless an entry exists
    for this exact shape AND was measured on the kernel sources being timed now."""
    try:
        with open(os.path.join(ROOT, "profiles", "r2_pmc_traffic.json")) as fh:
            tab = json.load(fh)
    except (OSError, ValueError):
        return None
    e = tab.get(f"{feats}x{patch}x{batch}x{dtype}")
    if not e or e.get("csrc_sha") != csrc_sha():
        return None
    return (2.0 * e["fetch_kib"] + e["write_kib"]) * 1024.0


def dominant_kernel_roofline(A, batch, patch, feats, dtype, iters=100):
    """The dominant kernel = the F->F 3x3 implicit-GEMM conv on [batch, patch, patch, F].  `achieved` / `frac` are
    quoted on its conv + bias + ReLU launch (the line's historical definition); `variants` times every flavour a
    training step issues -- residual epilogue, ReLU-mask (data-gradient) epilogue, weight gradient incl. its share of
    the grouped finalize -- and `step_weighted_frac` weights them by their launch counts in one ResBlock
    (forward: ReLU conv + residual conv; backward: mask dgrad + residual dgrad + 2 weight gradients)."""
    dt = TDT[dtype]
    dev = torch.device("cuda", torch.cuda.current_device())
    x = (torch.rand(batch, patch, patch, feats, device=dev) - 0.5).to(dt)
    x2 = (torch.rand(batch, patch, patch, feats, device=dev) - 0.5).to(dt)
    w = torch.nn.Parameter((torch.rand(feats, feats, 3, 3, device=dev) - 0.5) * 0.05)
    b = torch.nn.Parameter(torch.zeros(feats, device=dev))
    pk = A.ops.pack_conv(w, b, dt)
    pkd = A.ops.pack_conv(w, None, dt, dgrad=True)
    out = torch.empty_like(x)
    kw = dict(N=batch, H=patch, W=patch, Cin=feats, Cout=feats, out=out)
    flops = 2.0 * batch * patch * patch * feats * feats * 9
    peak = PEAK_TFLOPS[dtype]
    esz = 4 if dtype == "f32" else 2
    variants = {}
    us = _time_replays(lambda: A.ops.conv_raw(x, pk, relu=True, **kw), iters)
    variants["conv_bias_relu"] = us
    if dtype != "f32":
        variants["conv_scale_residual"] = _time_replays(lambda: A.ops.conv_raw(x, pk, scale=0.1, res=x2, **kw), iters)
        variants["dgrad_relu_mask"] = _time_replays(lambda: A.ops.conv_raw(x, pkd, mask=x2, use_bias=False, **kw), iters)
        # weight gradient the way a step runs it: 8 layers queued, one grouped launch + one grouped finalize
        ws = [torch.nn.Parameter(torch.zeros(feats, feats, 3, 3, device=dev)) for _ in range(8)]
        bs = [torch.nn.Parameter(torch.zeros(feats, device=dev)) for _ in range(8)]

        def wg():
            with A.ops.hold_wgrads():
                for wi, bi in zip(ws, bs):
                    A.ops.wgrad(x, x2, wparam=wi, bparam=bi, N=batch, H=patch, W=patch, Cin=feats, Cout=feats, k=3,
                                w_shape=(feats, feats, 3, 3))
        variants["wgrad_grouped_per_layer"] = _time_replays(wg, max(4, iters // 8)) / 8.0
    ach = flops / (us * 1e-6) / 1e12
    alg_bytes = 2.0 * batch * patch * patch * feats * esz            # one read + one write of the activation
    r = {"bound": "mfma", "kernel": f"conv_ws_kernel 3x3 {feats}->{feats} @{patch}x{patch} x{batch} ({dtype}), fwd = dgrad kernel",
         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
         "us_per_launch": round(us, 2), "flops_per_launch": flops,
         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps": round(alg_bytes / (us * 1e-6) / 1e9, 1),
         "traffic": pmc_traffic(feats, patch, batch, dtype),
         "variants_us": {k: round(v, 2) for k, v in variants.items()}}
    if len(variants) == 4:
        t = (variants["conv_bias_relu"] + 2 * variants["conv_scale_residual"] + variants["dgrad_relu_mask"]
             + 2 * variants["wgrad_grouped_per_layer"])
        r["step_weighted_frac"] = round(6 * flops / (t * 1e-6) / 1e12 / peak, 4)
    return r


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(a)                                        # never returns
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
    ddp = (world > 1 or force_ddp) and not a.inference
    use_torch_ddp = ddp and os.environ.get("SRK_USE_TORCH_DDP") == "1"
    # hipGraph modes.  N = 1: the whole step is ONE graph.  N > 1 (default, "segmented"): forward + loss + backward +
    # gradient packing are one graph, the bucketed all-reduce (RCCL over xGMI) is issued eagerly, the optimizer step is a
    # second graph -- no collective is ever inside a capture, so the N > 1 run keeps the graph's launch rate without
    # depending on RCCL's capture support.  SRK_BENCH_GRAPH_DDP=1: everything incl. the all-reduce in one graph (works on
    # a 1-rank group; opt-in).  --no-graph or a failed capture: eager launches, all-reduces overlapped with backward
    # by GradSync's hooks.
    mode = "eager" if a.no_graph else ("full" if (not ddp or os.environ.get("SRK_BENCH_GRAPH_DDP") == "1") else "segmented")
    if use_torch_ddp and mode == "segmented":
        mode = "eager"
    side = torch.cuda.Stream() if mode != "eager" else None
    if side is not None:
        side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
        net = T.wrap_ddp(model, dev, force=force_ddp) if use_torch_ddp else model
        gsync = None
        if ddp and not use_torch_ddp:
            gsync = T.GradSync(model, overlap=(mode == "eager"))
            gsync.broadcast()
        # reference optimizer: Adam at torch defaults (srmodel.py:145-154,602-603) = what configure_optimizers() returns:
        # torch.optim.Adam's update rule as ONE launch over all parameter tensors (optim.py / csrc/optim.hip), the step
        # counter on the device for hipGraph replay.  SRK_TORCH_ADAM=1: torch's own fused multi-tensor kernels (A/B).
        if os.environ.get("SRK_TORCH_ADAM") == "1":
            opt = torch.optim.Adam(params, fused=True, capturable=(mode != "eager"))
        else:
            opt = A.optim.Adam(params)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        sr = net(batch["lr"])
        loss = model._calculate_losses(img_sr=sr, img_hr=batch["hr"])["loss"]
        loss.backward()
        return loss

    def train_step():
        loss = fwd_bwd()
        if gsync is not None:
            gsync.sync()
        opt.step()
        return loss

    def infer_step():
        with torch.no_grad():
            return net(batch["lr"])

    step = infer_step if a.inference else train_step
    last = {}
    graphs = None
    if mode != "eager":
        try:
            with torch.cuda.stream(side):
                for _ in range(11 if ddp else 3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            cem = "thread_local" if ddp else "global"     # RCCL's watchdog thread polls events while this thread captures
            if mode == "full":
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side, capture_error_mode=cem):
                    last["out"] = step()
                graphs = (g,)
            else:
                ga, gb = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(ga, stream=side, capture_error_mode=cem):
                    last["out"] = fwd_bwd()
                    gsync.pack()
                with torch.cuda.graph(gb, stream=side, capture_error_mode=cem):
                    opt.step()
                graphs = (ga, gb)
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] hipGraph capture ({mode}) failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graphs, mode = None, "eager"
            torch.cuda.synchronize()

    def run_one():
        if graphs is None:
            last["out"] = step()
        elif len(graphs) == 1:
            graphs[0].replay()
        else:
            graphs[0].replay()
            gsync.reduce()
            graphs[1].replay()

    def timed(nsteps):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nsteps):
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
        return el

    for _ in range(a.warmup):
        run_one()
    torch.cuda.synchronize()
    loss_first = float(last["out"].detach().float().mean()) if not a.inference and "out" in last else None
    el = timed(a.steps)
    loss_last = float(last["out"].detach().float().mean()) if not a.inference and "out" in last else None
    # sustained rate: the chip lowers its clock under this kernel within ~1 s; the driver's K steps
    # are a burst.  The number of extra steps is fixed from the timed rate so that every rank runs the same count.
    sustained = None
    if a.sustain_seconds > 0:
        ns = max(a.steps, int(a.sustain_seconds / max(el / a.steps, 1e-6)))
        if world > 1:
            t = torch.tensor([ns], device=dev, dtype=torch.int64)
            dist.broadcast(t, src=0)
            ns = int(t.item())
        sustained = timed(ns) / ns

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
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "hip_graph": mode if graphs else False,
                       "grad_sync": (None if not ddp else "torch_ddp" if use_torch_ddp else "bucketed_allreduce"),
                       "loss_after_warmup": loss_first, "loss_after_timed_steps": loss_last},
            "model_mfma_frac": round(value / world * flop_per_patch / 1e3 / PEAK_TFLOPS[a.dtype], 4),
        }
        if sustained is not None:
            out["sustained_ms_per_step"] = round(sustained * 1e3, 4)
            out["sustained_value"] = round(a.batch * world / sustained, 2)
        if not a.no_roofline:
            try:
                out["roofline"] = dominant_kernel_roofline(A, a.batch, a.patch, feats, a.dtype)
            except Exception as e:  # noqa: BLE001
                out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(a.model, a.scale, a.patch, model=model, lr=batch["lr"], hr=batch["hr"])
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
