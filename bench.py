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
    p.add_argument("--no-other-configs", action="store_true",
                   help="skip the batch-16 runs of the five BASELINE models that follow the timed region of the default line")
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
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    out = {"value": round(n * k / dt, 3), "unit": "LR patches/s", "cores": cores, "cores_available": avail, "cpu_model": cpu_model, "kind": "port",
           "sample": f"{k} training steps of batch {n} ({cls} fp32, torch {torch.__version__} CPU, {cores} threads), {dt:.1f} s"}
    # SURVEY.md 8(d) defines the baseline on os.cpu_count() threads; the quoted figure uses min(32, cores) because torch's CPU convs collapse beyond that
    # on this host: ONE step on all 256 threads takes 107 s at batch 16 (0.15 patches/s) and 96 s at batch 1 (profiles/r6_bench_default_first.json; the
    # batch-1 figure: profiles/r6_experiments.txt 11) -- oversubscription, not work.  The all-cores step therefore runs on request only (SRK_BENCH_ALLCORES=1).
    out["all_cores"] = {"note": f"not measured in this run (SRK_BENCH_ALLCORES=1 adds one batch-1 step on all {avail} threads: ~96 s on this host); "
                                "measured once: 0.15 patches/s at batch 16 on 256 threads (profiles/r6_bench_default_first.json)"}
    if avail > cores and os.environ.get("SRK_BENCH_ALLCORES") == "1":
        try:
            torch.set_num_threads(avail)
            one = {"lr": batch["lr"][:1], "hr": batch["hr"][:1]}
            t1 = time.perf_counter()
            opt.zero_grad()
            m.training_step(one)["loss"].backward()
            opt.step()
            d1 = time.perf_counter() - t1
            out["all_cores"] = {"value": round(1 / d1, 3), "cores": avail, "sample": f"1 training step of batch 1 on {avail} threads, {d1:.1f} s"}
        finally:
            torch.set_num_threads(cores)
    if model is not None and lr is not None:
        try:
            import math
            np_ = min(2, lr.shape[0])
            with torch.no_grad():
                y_hip = model(lr[:np_]).float().cpu()
                m.load_state_dict({k_: v.detach().float().cpu() for k_, v in model.state_dict().items()})
                y_ref = m.forward(lr[:np_].float().cpu())
            mse = float(((y_hip - y_ref) ** 2).mean())
            out["parity"] = {"bench_weights": {"patches": np_, "psnr_build_vs_oracle_db": round(10.0 * math.log10(1.0 / max(mse, 1e-20)), 2),
                                               "max_abs_err": float((y_hip - y_ref).abs().max()),
                                               "note": "HIP forward in the bench dtype vs fp32 CPU oracle, the benched model's current weights, "
                                                       "synthetic (uniform) patches: a sanity check of the build, not the 0.01 dB criterion"}}
            if model_name == "edsr_baseline" and scale == 4 and os.environ.get("SRK_BENCH_NO_TRAINED_PARITY") != "1":
                import sr_amd as A_
                out["parity"]["trained_net"] = trained_parity(A_)
        except Exception as e:  # noqa: BLE001
            out["parity"] = {"error": f"{type(e).__name__}: {e}"}
    return out


def trained_parity(A, steps=200):
    """north_star's "PSNR within 0.01 dB of reference" where it means something: EDSR-baseline x4 TRAINED here (bf16 HIP path, Adam, `steps`
    steps on smooth synthetic images: oracle/images.py), then the same weights forward through (a) the fp32 CPU oracle = the reference path,
    (b) the HIP model in bf16 storage, (c) in fp16 storage, (d) `predict_step` of the bf16 model (its `eval_dtype`: fp16 storage with a bf16
    fall-back on overflow).  delta = PSNR(build, hr) - PSNR(oracle, hr), data-set mean and worst single image.  Part of the cpu_baseline leg
    (the only part of the bench that may touch oracle/); the same procedure as tests/test_gpu_fullsize_parity.py::
    test_psnr_within_0p01_db_of_reference_path (300 steps, 6 images there)."""
    import torch.nn.functional as F
    from oracle import functional as OF
    from oracle.images import psnr, smooth_images
    kw = dict(MODELS["edsr_baseline"][1])
    torch.manual_seed(0)
    m = A.EDSR(precision="bf16", scale_factor=4, **kw).cuda()
    hr = smooth_images(32, 192, 11)
    lr = F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    hr_d, lr_d = hr.cuda(), lr.cuda()
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-3)
    g = torch.Generator().manual_seed(1)
    first = last = None
    for step in range(steps):
        idx = torch.randint(0, hr.shape[0], (16,), generator=g).cuda()
        opt.zero_grad(set_to_none=True)
        loss = m.training_step({"lr": lr_d[idx], "hr": hr_d[idx]}, step)["loss"]
        loss.backward()
        opt.step()
        first = float(loss.detach()) if first is None else first
        last = float(loss.detach())
    sd = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    hr_t = smooth_images(4, 192, 99)
    lr_t = F.interpolate(hr_t, scale_factor=0.25, mode="bicubic", antialias=True).clamp(0, 1)
    with torch.no_grad():
        y_ref = OF.forward("EDSR", sd, lr_t, scale_factor=4, **kw)
    p_ref = psnr(y_ref, hr_t)
    out = {"net": f"EDSR-baseline x4 trained {steps} Adam steps on the HIP bf16 path (L1 {first:.4f} -> {last:.4f}), 4 smooth 48x48 test patches",
           "psnr_reference_path_db": round(p_ref, 3), "criterion_db": 0.01, "delta_psnr_db": {}, "worst_image_delta_db": {}}

    def deltas(y):
        per = [psnr(y[i:i + 1], hr_t[i:i + 1]) - psnr(y_ref[i:i + 1], hr_t[i:i + 1]) for i in range(hr_t.shape[0])]
        return round(psnr(y, hr_t) - p_ref, 4), round(max(per, key=abs), 4)
    for name, prec in (("bf16", "bf16"), ("fp16", 16)):
        m2 = A.EDSR(precision=prec, scale_factor=4, **kw)
        m2.load_state_dict(sd)
        m2 = m2.cuda().eval()
        with torch.no_grad():
            out["delta_psnr_db"][name], out["worst_image_delta_db"][name] = deltas(m2(lr_t.cuda()).float().cpu())
            if name == "bf16":
                ye = m2.predict_step({"lr": lr_t.cuda()}, 0).float().cpu()
                out["delta_psnr_db"]["bf16_model_eval_dtype_path"], out["worst_image_delta_db"]["bf16_model_eval_dtype_path"] = deltas(ye)
                out["eval_dtype"] = str(m2.eval_dtype)
    out["note"] = ("a RAW bf16 forward (8 mantissa bits on the 16-block trunk) sits 0.005-0.013 dB below the reference path; validation_step / "
                   "predict_step of a bf16 model therefore run in fp16 storage (SRModel.eval_dtype) -- config 2's 0.01 dB is met by that path, "
                   "not by the bf16 kernels alone")
    return out


def _time_replays(fn, iters, sustain_s=0.0):
    """Average duration (us) of one call of `fn`: `iters` calls are captured into one hipGraph and replayed between two
    HIP events on the launch stream, so the Python launch rate (~10 us) does not enter.  Returns (burst, sustained):
    burst = one replay right after a warm-up replay; sustained = the replays of the second half of `sustain_s` seconds of
    back-to-back replays (the chip lowers its clock under these kernels within ~1 s: DESIGN.md section 7), None if 0."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
    torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    import sr_amd as A
    g = torch.cuda.CUDAGraph()
    with A.ops.graph_capture(g):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    burst = e0.elapsed_time(e1) * 1e3 / iters
    if sustain_s <= 0:
        return burst, None
    nrep = max(4, int(sustain_s * 1e6 / max(burst * iters, 1.0)))
    for _ in range(nrep // 2):
        g.replay()
    e0.record()
    for _ in range(nrep - nrep // 2):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return burst, e0.elapsed_time(e1) * 1e3 / (iters * (nrep - nrep // 2))


def kernel_fingerprint(key="conv_ws_plain_bf16"):
    """Fingerprint of the GENERATED ISA of one kernel of the library being timed (csrc/kernel_isa.json, written by every build:
    csrc/check_isa.py hashes the kernel's own instructions, so an edit elsewhere in the sources -- another kernel, a prototype in
    a shared header -- does not orphan a PMC measurement; round 3 hashed whole source files and lost `roofline.traffic` to one)."""
    try:
        with open(os.path.join(ROOT, "sr-pytorch-lightning_amd", "csrc", "kernel_isa.json")) as fh:
            return json.load(fh).get(key)
    except (OSError, ValueError):
        return None


def pmc_traffic(key):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE,
    KiB -> B; MI355X_MICROARCH.md "HBM"), written by tools/pmc_traffic.sh.  None unless an entry exists for this exact
    kernel / shape AND was measured on the very instructions being timed now (`isa_sha` == this build's fingerprint)."""
    for tag in ("r6", "r5", "r4"):           # the newest collection whose fingerprint matches the instructions being timed
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")) as fh:
                tab = json.load(fh)
        except (OSError, ValueError):
            continue
        e = tab.get(key)
        if e and e.get("isa_sha") and e.get("isa_sha") == kernel_fingerprint(e.get("isa_key", "conv_ws_plain_bf16")):
            return (2.0 * e["fetch_kib"] + e["write_kib"]) * 1024.0
    return None


def pmc_step_entry(model_name, batch, dtype):
    """The dominant kernel of a configuration INSIDE its training step under the PMC passes of tools/pmc_traffic.sh (`step:<model>:b<batch>:<dtype>`):
    {kernel, hbm_bytes_per_launch, algorithmic_bytes_per_launch, ...}, or None unless it was measured on the instructions being timed."""
    for tag in ("r6",):
        try:
            with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")) as fh:
                e = json.load(fh).get(f"step:{model_name}:b{batch}:{dtype}")
        except (OSError, ValueError):
            continue
        if e and e.get("isa_sha") and e["isa_sha"] == kernel_fingerprint(e.get("isa_key")):
            return e
    return None


def _flavours(A, model_name, batch, patch, feats, dtype):
    """The launches that carry the model's residual blocks at this batch: (kernel name, [flavour dicts]).  A flavour =
    name, fn (one launch, or `per` layers' worth of launches), flops per layer (ALGORITHMIC: 2 x MACs of the reference
    convs, no padding / halo / recompute work), count = launches of it per block (forward + backward)."""
    ops = A.ops
    dt = TDT[dtype]
    dev = torch.device("cuda", torch.cuda.current_device())
    n, hw = batch, patch

    def act(c):
        return (torch.rand(n, hw, hw, c, device=dev) - 0.5).to(dt)

    def par(co, ci, k):
        return torch.nn.Parameter((torch.rand(co, ci, k, k, device=dev) - 0.5) * (1.0 / (ci * k * k) ** 0.5)), torch.nn.Parameter(torch.zeros(co, device=dev))

    def wg_flavour(xin, dy, cin, cout, k, flops, count):
        ws = [par(cout, cin, k) for _ in range(8)]

        def wg():
            with ops.hold_wgrads():
                for wi, bi in ws:
                    ops.wgrad(xin, dy, wparam=wi, bparam=bi, N=n, H=hw, W=hw, Cin=xin.shape[3], Cout=dy.shape[3], k=k, w_shape=(cout, cin, k, k))
        return dict(name="wgrad_grouped_per_layer", fn=wg, flops=flops, count=count, per=8)

    px = n * hw * hw
    if model_name == "wdsr_b" and dtype != "f32":
        f, chid, cmid = 128, 768, 102
        x, g, z, gz = act(f), act(f), act(112), act(112)
        gz[..., cmid:] = 0
        (w1, b1), (w2, b2), (w3, b3) = par(chid, f, 1), par(cmid, chid, 1), par(f, cmid, 3)
        pk = ops.pw_pack(w1, b1, w2, b2, dt)
        pk3, pk3d = ops.pack_conv(w3, b3, dt), ops.pack_conv(w3, None, dt, dgrad=True)
        out, gx = torch.empty_like(x), torch.empty_like(x)
        fpw, f3 = 2.0 * px * chid * (f + cmid), 2.0 * px * cmid * f * 9
        fl = [dict(name="pw_pair_fwd", fn=lambda: ops.pw_forward_raw(x, pk, z), flops=fpw, count=1),
              dict(name="pw_pair_bwd_data", fn=lambda: ops.pw_backward_raw(x, gz, pk, gx, res=g), flops=fpw, count=1),
              dict(name="pw_pair_wgrad", fn=lambda: ops.pw_wgrad_raw(x, gz, pk, tuple(w1.shape), tuple(w2.shape)), flops=fpw, count=1),
              dict(name="conv3x3_scale_residual", fn=lambda: ops.conv_raw(z, pk3, N=n, H=hw, W=hw, Cin=112, Cout=f, out=out, scale=1.0, res=x), flops=f3, count=1),
              dict(name="dgrad3x3", fn=lambda: ops.conv_raw(g, pk3d, N=n, H=hw, W=hw, Cin=f, Cout=112, out=gz, use_bias=False), flops=f3, count=1),
              wg_flavour(z, g, cmid, f, 3, f3, 1)]
        return "pw_fwd_kernel + pw_bwd_kernel + pw_wgrad_kernel (WDSR-B pointwise pair 128->768->102, csrc/pw_chain.hip) and conv_ks_kernel 3x3 102->128", fl
    if model_name == "ddbpn" and dtype != "f32" and patch % 4 == 0:
        # D-DBPN's projection convs (ddbpn.py:10-24: kernel 8, stride 4, padding 2, 32 <-> 32 channels): 33 per step, each once as
        # `up` or `down` forward, once as the other for its data gradient, once as weight gradient (csrc/proj.hip).  HBM-bound:
        # 4.8 GFLOP per launch against the 32-channel HR tensor (algorithmic bytes = HR tensor + LR tensor, read or written once)
        import ctypes as C
        lib = A._lib.load()
        xl = act(32)
        xh = (torch.rand(n, 4 * hw, 4 * hw, 32, device=dev) - 0.5).to(dt)
        w4 = (torch.rand(32, 32, 8, 8, device=dev) - 0.5) * 0.05
        b32 = torch.zeros(32, device=dev)
        half = lib.srk_proj_pack_bytes() // 2
        wpk = torch.empty(2 * half, dtype=torch.uint8, device=dev)
        A._lib.check(lib.srk_proj_pack(w4.data_ptr(), wpk.data_ptr(), ops._DT[dt], torch.cuda.current_stream().cuda_stream), "srk_proj_pack")
        dw, db = torch.empty(32, 32, 8, 8, device=dev), torch.empty(32, device=dev)
        scratch = torch.empty(lib.srk_proj_wgrad_scratch_floats(n, hw, hw), dtype=torch.float32, device=dev)

        def wg():
            A._lib.call("srk_proj_wgrad", A._lib.ProjWgradArgs(xh=xh.data_ptr(), xh_pitch=32, g=xl.data_ptr(), g_pitch=32, scratch=scratch.data_ptr(),
                                                             dw=dw.data_ptr(), accumulate=0, N=n, H=hw, W=hw, dtype=ops._DT[dt], db=db.data_ptr(),
                                                             bias_side=2, db_accumulate=0), torch.cuda.current_stream().cuda_stream)
        fpr = 2.0 * px * 2048 * 32
        by = float(17 * px * 32 * 2)
        fl = [dict(name="proj_up", fn=lambda: ops._proj_launch(xl, wpk[half:], b32, True), flops=fpr, bytes=by, count=1),
              dict(name="proj_down", fn=lambda: ops._proj_launch(xh, wpk[:half], b32, False), flops=fpr, bytes=by, count=1),
              dict(name="proj_wgrad", fn=wg, flops=fpr, bytes=by, count=1)]
        return "proj_up_kernel / proj_down_kernel / proj_wgrad_kernel (D-DBPN projection 8x8 stride 4, 32 <-> 32 channels, csrc/proj.hip)", fl
    if model_name == "rdn_b" and dtype != "f32":
        cin, g0 = 320, 64                  # the middle dense layer of an RDB (rdn.py:9-21: Cin = 64 + 64 c)
        x, y, gy = act(cin), act(g0), act(g0)
        w, b = par(g0, cin, 3)
        pkf, pkd = ops.pack_conv(w, b, dt), ops.pack_conv(w, None, dt, dgrad=True)
        gx = torch.empty_like(x)
        fl0 = 2.0 * px * cin * g0 * 9
        fl = [dict(name="dense_conv_relu", fn=lambda: ops.conv_raw(x, pkf, N=n, H=hw, W=hw, Cin=cin, Cout=g0, out=y, relu=True), flops=fl0, count=1),
              dict(name="dense_dgrad_accumulate", fn=lambda: ops.conv_raw(gy, pkd, N=n, H=hw, W=hw, Cin=g0, Cout=cin, out=gx, res=gx, use_bias=False), flops=fl0, count=1),
              wg_flavour(x, gy, cin, g0, 3, fl0, 1)]
        return f"conv_ks_kernel 3x3 {cin}->{g0} (RDN dense layer)", fl
    x, x2 = act(feats), act(feats)
    out = torch.empty_like(x)
    w, b = par(feats, feats, 3)
    flops = 2.0 * px * feats * feats * 9
    pk = ops.pack_conv(w, b, dt)
    if dtype == "f32":
        kwc = dict(N=n, H=hw, W=hw, Cin=feats, Cout=feats, out=out)
        return f"conv_igemm_kernel 3x3 {feats}->{feats} (fp32 MFMA)", [dict(name="conv_bias_relu", fn=lambda: ops.conv_raw(x, pk, relu=True, **kwc), flops=flops, count=1)]
    pkd = ops.pack_conv(w, None, dt, dgrad=True)
    if feats == 64 and model_name != "srresnet" and ops.pair_ok(x, w, w):      # (SRResNet: a BatchNorm sits between the two convs of a block -- single conv_ws launches)
        # the reference's batch: two convs per launch (csrc/conv_pair.hip); RCAN: the CALayer steps ride on the launches
        w2, b2 = par(feats, feats, 3)
        pk2, pk2d = ops.pack_conv(w2, b2, dt), ops.pack_conv(w2, None, dt, dgrad=True)
        mid, g1 = torch.empty_like(x), torch.empty_like(x)
        if model_name == "rcan":
            lib = A._lib.load()
            ns, cr = lib.srk_conv_pair_tiles(1, hw, hw), 4
            f32 = dict(dtype=torch.float32, device=dev)
            sums, sums2, gsum = torch.rand(n, ns, 64, **f32), torch.empty(n, ns, 64, **f32), torch.rand(n, ns, 64, **f32)
            cw1, cb1, cw2, cb2 = torch.rand(cr, 64, **f32) * 0.1, torch.zeros(cr, **f32), torch.rand(64, cr, **f32) * 0.1, torch.zeros(64, **f32)
            s_, z_ = torch.rand(n, 64, **f32), torch.rand(n, cr, **f32)
            per = torch.empty(n, 2 * cr * 64 + cr + 64, **f32)
            xo, t = torch.empty_like(x), act(64)
            fwd = lambda: ops.conv_pair_raw(x, pk, pk2, out=out, relu_mid=True, mid=mid, pool=sums2, xo=xo,
                                            ca_fwd=dict(x2=x2, sums=sums, w1=cw1, b1=cb1, w2=cw2, b2=cb2, s_out=s_, z_out=z_))
            bwd = lambda: ops.conv_pair_raw(x, pk2d, pkd, out=out, mask=x2, mid=g1, res=x, use_bias=False, pool=sums2, pool_aux=t, xo=xo,
                                            ca_bwd=dict(gsum=gsum, sums=sums, s=s_, z=z_, w1=cw1, w2=cw2, slots=per))
            names = ("pair_rcab_fwd_ca_in_pool", "pair_rcab_bwd_ca_in_pool")
        else:
            fwd = lambda: ops.conv_pair_raw(x, pk, pk2, out=out, relu_mid=True, mid=mid, scale_out=0.1, res=x)
            bwd = lambda: ops.conv_pair_raw(x, pk2d, pkd, out=out, scale_mid=0.1, mask=x2, mid=g1, res=x, use_bias=False)
            names = ("pair_resblock_fwd", "pair_resblock_bwd")
        fl = [dict(name=names[0], fn=fwd, flops=2 * flops, count=1), dict(name=names[1], fn=bwd, flops=2 * flops, count=1),
              wg_flavour(x, x2, feats, feats, 3, flops, 2)]
        return f"conv_pair_kernel: two 3x3 {feats}->{feats} convs per launch (csrc/conv_pair.hip)", fl
    kwc = dict(N=n, H=hw, W=hw, Cin=feats, Cout=feats, out=out)
    # the launches a ResBlock issues in a TRAINING step (ops.ConvChainFn): conv + bias + ReLU also writes the ReLU sign bits (4 bytes per
    # pixel and 32-channel half), the data gradient behind the ReLU masks with them (the 16-bit activation `mask=` is the fallback for
    # kernels without the bit path: conv_raw picks)
    ops.conv_raw(x, pk, relu=True, relu_bits="want", **kwc)
    bits = out.__dict__.pop("_srk_bits", None)

    def fwd_relu():
        ops.conv_raw(x, pk, relu=True, relu_bits="want", **kwc)
        out.__dict__.pop("_srk_bits", None)
    fl = [dict(name="conv_bias_relu", fn=fwd_relu, flops=flops, count=1),
          dict(name="conv_scale_residual", fn=lambda: ops.conv_raw(x, pk, scale=0.1, res=x2, **kwc), flops=flops, count=2),
          dict(name="dgrad_relu_mask", fn=lambda: ops.conv_raw(x, pkd, mask=x2, mask_bits=bits, scale=0.1, use_bias=False, **kwc), flops=flops, count=1),
          wg_flavour(x, x2, feats, feats, 3, flops, 2)]
    kern = "conv_ws_kernel" if feats == 64 else "conv_ks_kernel"
    return f"{kern} 3x3 {feats}->{feats}, fwd = dgrad kernel", fl


def _replay_us(capture, sustain_s):
    """`capture()` issues launches; they are captured into ONE hipGraph, replayed back to back for `sustain_s` seconds, and the
    second half of the replays is timed by HIP events on the launch stream.  Returns microseconds per replay."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            capture()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    import sr_amd as A
    g = torch.cuda.CUDAGraph()
    with A.ops.graph_capture(g, stream=side):
        capture()
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    one = max(e0.elapsed_time(e1) * 1e3, 1.0)
    nrep = max(4, int(sustain_s * 1e6 / one))
    for _ in range(nrep // 2):
        g.replay()
    e0.record()
    for _ in range(nrep - nrep // 2):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (nrep - nrep // 2)
    del g
    return us


def _uses_trunk(ops, model, f0):
    blocks = list(model.body)[:-1]
    specs = [blk.plain_convs() if hasattr(blk, "plain_convs") else None for blk in blocks]
    tail = model.body[-1]
    return all(sp is not None for sp in specs) and hasattr(ops, "res_trunk_ok") and bool(ops.res_trunk_ok(f0, specs, (tail.weight, tail.bias)))


def in_step_body(A, model, batch, patch, dtype, sustain_s=0.6):
    """The dominant kernel family WHERE IT LIVES: the residual trunk of the benched model -- `model.body_nhwc`,
    the code its forward runs -- as it runs inside a training step: autograd recording (the conv + ReLU launches store their sign
    bits, every layer writes its own activation buffer: 33 x 75 MB at batch 256, nothing is re-read from the Infinity Cache the way
    an isolated launch re-reads its one input), the flavours alternating in dependency order, the data gradients behind them in
    the backward pass's order, sustained clocks.  Three graphs, replayed back to back and timed by HIP events on the launch stream:
        F   = weight packing + trunk forward                       (n convs)
        FB  = F + backward with the weight gradients left queued   (+ n data-gradient launches)
        FBW = F + backward + the grouped weight-gradient launch and its finalize
    and P = the packing launch alone.  in-step time per convolution of the family = (FB - P) / 2n; algorithmic FLOPs = 2 x F x F x 9
    per pixel and convolution.  (The reference has no counterpart: cuDNN picks its kernels.)  Returns None for models without `body_nhwc`."""
    if not hasattr(model, "body_nhwc") or dtype == "f32":
        return None
    ops = A.ops
    n_conv, feats = model.body_conv_launches()
    dev = torch.device("cuda", torch.cuda.current_device())
    dt = TDT[dtype]
    f0 = (torch.rand(batch, patch, patch, feats, device=dev) - 0.5).to(dt)
    g0 = ((torch.rand(batch, patch, patch, feats, device=dev) - 0.5) * 1e-3).to(dt)
    params = [p for p in model.body.parameters() if p.requires_grad]

    def pack_only():
        with ops.forward_scope(model._pack_group()):
            pass

    def fwd():
        with ops.forward_scope(model._pack_group()):
            f = f0.detach().requires_grad_(True)
            return f, model.body_nhwc(f)

    def fwd_only():
        fwd()

    def fwd_bwd(flush):
        for p in params:
            p.grad = None
        f, r = fwd()
        if flush:
            torch.autograd.backward(r, g0)
        else:
            with ops.hold_wgrads_discard():
                torch.autograd.backward(r, g0)

    t_p = _replay_us(pack_only, 0.05)
    t_f = _replay_us(fwd_only, sustain_s)
    t_fb = _replay_us(lambda: fwd_bwd(False), sustain_s)
    t_fbw = _replay_us(lambda: fwd_bwd(True), sustain_s)
    for p in params:
        p.grad = None
    px = batch * patch * patch
    flops = 2.0 * px * feats * feats * 9
    fam_us = (t_fb - t_p) / (2 * n_conv)
    # at small batches (ops.pair_ok) one launch carries TWO of the trunk's convolutions (csrc/conv_pair.hip): per-launch = 2 x per-conv
    wprobe = torch.empty(feats, feats, 3, 3, device=dev)
    cpl = 2 if (feats == 64 and ops.pair_ok(f0, wprobe, wprobe)) else 1
    # a batch of whole rounds over the CUs (ops.res_trunk_ok): the trunk is ONE image-stationary launch per direction (csrc/conv_igemm.hip,
    # conv_trunk_kernel: n_conv layers per launch, + the long skip's elementwise add in the backward one) -- F and FB - F are one launch each
    trunk = type(model).__name__ == "EDSR" and feats == 64 and _uses_trunk(ops, model, f0)
    if trunk:
        cpl = n_conv
    # the same launches against the OTHER roofline: a 64 -> 64 3x3 convolution on bf16 sits at the ridge (2 x 64 x 64 x 9 FLOP over 2 x 64 x 2
    # bytes per pixel = 288 FLOP/B against 2.5 PFLOP/s / 8 TB/s = 312), and the flavours that also read a residual (every second forward
    # launch, every second data gradient) below it: 192 FLOP/B.  EDSR's trunk: per ResBlock forward 2 + 3 activation tensors, backward
    # 2 (+ the sign bits, 1/16) + 3, the trunk's last convolution 3 forward (long skip) and 2 backward.
    hbm = None
    if type(model).__name__ == "EDSR" and (cpl == 1 or trunk):
        T = px * feats * 2.0
        nb = (n_conv - 1) // 2
        tot_b = (nb * 5 + 3) * T + (nb * (5 + 1.0 / 16) + 2) * T
        per = tot_b / (2 * n_conv)
        hbm = {"algorithmic_bytes_per_conv": round(per), "achieved_GBps": round(per / (fam_us * 1e-6) / 1e9, 1), "peak_GBps": 8000.0,
               "frac": round(per / (fam_us * 1e-6) / 1e9 / 8000.0, 4), "flop_per_byte": round(flops / per, 1), "ridge_flop_per_byte": 312.5,
               "note": "activation tensors each launch must read and write once (input, output, residual where the flavour has one, sign bits), "
                       "averaged over the trunk's forward + data-gradient launches; below the ridge, so by the roofline model these launches are "
                       "bound by HBM, not by the matrix pipes"}
    return {"convs": 2 * n_conv, "convs_per_launch": cpl, "launch": "conv_trunk_kernel" if trunk else None, "us_per_conv": round(fam_us, 2), "us_per_launch": round(fam_us * cpl, 2), "hbm_view": hbm,
            "flops_per_launch": flops * cpl,
            "fwd_us_per_conv": round((t_f - t_p) / n_conv, 2), "dgrad_us_per_conv": round((t_fb - t_f) / n_conv, 2),
            "wgrad_us_per_layer": round((t_fbw - t_fb) / n_conv, 2),
            "graph_us": {"pack": round(t_p, 1), "fwd": round(t_f, 1), "fwd_bwd": round(t_fb, 1), "fwd_bwd_wgrad": round(t_fbw, 1)},
            "frac": round(flops / (fam_us * 1e-6) / 1e12 / PEAK_TFLOPS[dtype], 4),
            "frac_with_wgrad": round(3 * n_conv * flops / ((t_fbw - t_p) * 1e-6) / 1e12 / PEAK_TFLOPS[dtype], 4),
            "method": "the model's own trunk (body_nhwc) captured as hipGraphs in training mode (autograd on, one buffer per layer), replayed "
                      f"for {sustain_s:g} s each, second half timed by HIP events on the launch stream; FB = forward + data gradients "
                      "(weight gradients queued, not launched), minus the weight-packing launch"}


def _attach_step_traffic(r, model_name, batch, dtype):
    """`traffic` = HBM-side bytes per launch of the quoted kernel as the PMC passes counted them IN THE STEP of this configuration
    (pmc_step_entry); the isolated-launch figure (pmc_traffic) stays where no in-step pass exists."""
    e = pmc_step_entry(model_name, batch, dtype)
    if e is None:
        return
    r["traffic"] = e["hbm_bytes_per_launch"]
    r["traffic_kernel"] = e["kernel"]
    if e.get("algorithmic_bytes_per_launch"):
        r["algorithmic_bytes_per_launch"] = e["algorithmic_bytes_per_launch"]
        r["algorithmic_bytes_are"] = e.get("algorithmic_bytes_are")
    if r.get("algorithmic_bytes_per_launch"):
        r["traffic_over_algorithmic"] = round(r["traffic"] / r["algorithmic_bytes_per_launch"], 3)


def dominant_kernel_roofline(A, model_name, batch, patch, feats, dtype, iters=100, sustain_s=1.0, in_step=None):
    """The kernel(s) that carry the model's residual blocks AT THIS BATCH (`_flavours`), each flavour a training step issues timed
    by HIP events on the launch stream: a burst (one graph of `iters` launches right after a warm-up replay) and SUSTAINED
    (>= `sustain_s` seconds of back-to-back replays, the second half timed: the clock has dropped by then, which is what a
    rocprofv3 run of the same launches sees -- profiles/r3_variants_*.txt).  `achieved` / `frac` / `variants_us` /
    `step_weighted_frac` are the SUSTAINED figures; the `_burst` twins are reported next to them.  step-weighted = the block's
    algorithmic FLOPs over the sum of its launches' times (forward + data gradient + weight gradients)."""
    kernel, fl = _flavours(A, model_name, batch, patch, feats, dtype)
    peak = PEAK_TFLOPS[dtype]
    esz = 4 if dtype == "f32" else 2
    burst, sust = {}, {}
    for f in fl:
        per = f.get("per", 1)
        b_, s_ = _time_replays(f["fn"], max(4, iters // per), sustain_s)
        burst[f["name"]], sust[f["name"]] = b_ / per, (s_ if s_ is not None else b_) / per
    f0 = fl[0]
    us = sust[f0["name"]]
    if "bytes" in f0:            # an HBM-bound path: algorithmic bytes per launch over the launch time, against ~8 TB/s
        ach = f0["bytes"] / (us * 1e-6) / 1e9
        r = {"bound": "hbm", "kernel": f"{kernel} @{patch}x{patch} x{batch} ({dtype}); quoted flavour: {f0['name']}",
             "achieved": round(ach, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 4),
             "frac_burst": round(f0["bytes"] / (burst[f0["name"]] * 1e-6) / 1e9 / 8000.0, 4),
             "us_per_launch": round(us, 2), "algorithmic_bytes_per_launch": f0["bytes"], "flops_per_launch": f0["flops"],
             "mfma_frac": round(f0["flops"] / (us * 1e-6) / 1e12 / peak, 4), "traffic": None,
             "variants_us": {k: round(v, 2) for k, v in sust.items()},
             "variants_us_burst": {k: round(v, 2) for k, v in burst.items()},
             "timing": f"HIP events around hipGraph replays; sustained = second half of {sustain_s:g} s of replays per flavour"}
        tot = sum(f["count"] * f["bytes"] for f in fl)
        r["step_weighted_frac"] = round(tot / (sum(f["count"] * sust[f["name"]] for f in fl) * 1e-6) / 1e9 / 8000.0, 4)
        r["launches_per_block"] = {f["name"]: f["count"] for f in fl}
        _attach_step_traffic(r, model_name, batch, dtype)
        return r
    ach = f0["flops"] / (us * 1e-6) / 1e12
    px = batch * patch * patch
    alg_bytes = 2.0 * px * feats * esz                              # one read + one write of the block's activation
    iso = {"quoted_flavour": f0["name"], "us_per_launch": round(us, 2), "achieved": round(ach, 2), "frac": round(ach / peak, 4),
           "frac_burst": round(f0["flops"] / (burst[f0["name"]] * 1e-6) / 1e12 / peak, 4),
           "variants_us": {k: round(v, 2) for k, v in sust.items()},
           "variants_us_burst": {k: round(v, 2) for k, v in burst.items()},
           "timing": f"each flavour ALONE on the chip, re-reading its own buffers: HIP events around hipGraph replays; sustained = second half of {sustain_s:g} s of replays"}
    if len(fl) > 1:
        tot = sum(f["count"] * f["flops"] for f in fl)
        iso["step_weighted_frac"] = round(tot / (sum(f["count"] * sust[f["name"]] for f in fl) * 1e-6) / 1e12 / peak, 4)
        iso["step_weighted_frac_burst"] = round(tot / (sum(f["count"] * burst[f["name"]] for f in fl) * 1e-6) / 1e12 / peak, 4)
        iso["launches_per_block"] = {f["name"]: f["count"] for f in fl}
    r = {"bound": "mfma", "kernel": f"{kernel} @{patch}x{patch} x{batch} ({dtype})",
         "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
         "us_per_launch": round(us, 2), "flops_per_launch": f0["flops"], "where": f"isolated launches of {f0['name']} (no in-step measurement for this model)",
         "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_GBps": round(alg_bytes / (us * 1e-6) / 1e9, 1),
         "traffic": pmc_traffic(f"{f0['name']}:{model_name}:{feats}x{patch}x{batch}x{dtype}"),
         "isolated": iso}
    if in_step is not None:
        # the ONE fraction of the line: the family's launches inside the step (in_step_body), executed = algorithmic FLOPs per launch
        ach = in_step["flops_per_launch"] / (in_step["us_per_launch"] * 1e-6) / 1e12
        if in_step.get("launch"):
            r["kernel"] = (f"{in_step['launch']}: the trunk's {in_step['convs_per_launch']} 3x3 {feats}->{feats} convolutions per direction in ONE image-stationary "
                           f"launch, each layer the body of conv_ws_kernel (csrc/conv_igemm.hip) @{patch}x{patch} x{batch} ({dtype})")
        r["algorithmic_bytes_per_launch"] = alg_bytes * in_step["convs_per_launch"]
        r.update({"achieved": round(ach, 2), "frac": round(ach / peak, 4), "us_per_launch": in_step["us_per_launch"],
                  "flops_per_launch": in_step["flops_per_launch"], "algorithmic_GBps": round(alg_bytes / (in_step["us_per_launch"] * 1e-6) / 1e9, 1),
                  "where": f"IN THE STEP: average over the {in_step['convs']} forward + data-gradient 3x3 convolutions of the trunk "
                           f"({in_step['convs_per_launch']} per launch) in a training step's order, buffers and autograd mode (`in_step`); "
                           "`isolated` = each flavour alone on the chip",
                  "in_step": in_step})
        if in_step.get("hbm_view"):
            r["algorithmic_bytes_per_launch"] = in_step["hbm_view"]["algorithmic_bytes_per_conv"] * in_step["convs_per_launch"]
    _attach_step_traffic(r, model_name, batch, dtype)
    return r


def executed_gflop_fwd(model_name, gflop_fwd, patch, scale, dtype, collapsed):
    """Conv GFLOP per patch the HIP path EXECUTES in a forward pass.  The reference's module graph ends EDSR / RCAN /
    RDN with conv3x3(64 -> 256) -> PixelShuffle(2) -> conv3x3(64 -> 3); on the 16-bit path those two layers run as ONE 5x5 convolution
    64 -> 12 at the lower resolution (ops.hr_tail, csrc/hr_tail.hip: same function and gradients, 8x fewer multiply-adds), so the executed
    count is lower than the reference graph's.  Both are reported; `model_mfma_frac` stays on the reference graph's count."""
    g = gflop_fwd * (patch / 48.0) ** 2
    if collapsed and dtype != "f32" and model_name in ("edsr_baseline", "rcan", "rdn_b") and scale in (2, 4):
        hw = patch * scale // 2                      # resolution of the last upsampler stage's input
        two_layers = 2.0 * hw * hw * (256 * 64 * 9 + 4 * 3 * 64 * 9) / 1e9
        one_conv = 2.0 * hw * hw * (12 * 64 * 25) / 1e9
        g = g - two_layers + one_conv
    return g


def quick_train_rate(A, T, name, batch, patch, scale, dtype, seconds=1.5, keep_model=False):
    """One more BASELINE config in this process (after the timed region): the training step of `name` at `batch` as one
    hipGraph, replayed for >= `seconds`; returns patches/s and the model-level MFMA fraction."""
    dev = torch.device("cuda", torch.cuda.current_device())
    cls, kw, gflop_fwd, _ = MODELS[name]
    torch.manual_seed(0)
    model = getattr(A, cls)(scale_factor=scale, precision=PREC[dtype], **kw).to(dev)
    batch_t = T.synthetic_batch(batch, 3, patch, scale, 4321, dev)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt = A.optim.Adam([p for p in model.parameters() if p.requires_grad])
        scaler = A.optim.DeviceGradScaler(dev) if dtype == "f16" else None      # fp16: device-resident dynamic loss scaling

        def step():
            opt.zero_grad(set_to_none=True)
            loss = model._calculate_losses(img_sr=model(batch_t["lr"]), img_hr=batch_t["hr"])["loss"]
            if scaler is None:
                A.ops.backward(loss)          # (loss.backward() with a cached seed gradient instead of a per-pass fill launch)
                opt.step()
            else:
                A.ops.backward(scaler.scale(loss))
                opt.step(grad_scaler=scaler)
            return loss
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with A.ops.graph_capture(g, stream=side):
        loss = step()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = 0
    while k < 10 or time.perf_counter() - t0 < seconds:
        for _ in range(5):
            g.replay()
        k += 5
        torch.cuda.synchronize()
    el = time.perf_counter() - t0
    value = batch * k / el
    flop = gflop_fwd * 3.0 * (patch / 48.0) ** 2
    flop_x = 3.0 * executed_gflop_fwd(name, gflop_fwd, patch, scale, dtype, A.ops._HR_COLLAPSE)
    out = {"model": name, "batch": batch, "value": round(value, 1), "unit": "LR patches/s", "ms_per_step": round(el / k * 1e3, 4), "steps": k,
           "model_mfma_frac": round(value * flop / 1e3 / PEAK_TFLOPS[dtype], 4),
           "model_mfma_frac_executed": round(value * flop_x / 1e3 / PEAK_TFLOPS[dtype], 4), "loss": float(loss.detach().float())}
    if scaler is not None:
        out["loss_scale"] = {"scale": scaler.get_scale(), "skipped_steps": scaler.skipped_steps}
    del g, opt, batch_t
    if keep_model:
        for p_ in model.parameters():
            p_.grad = None
        torch.cuda.empty_cache()
        return out, model
    del model
    torch.cuda.empty_cache()
    return out


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
    if world > 1:
        # one process per GPU means one GPU per process: every rank reports the device it computes on and all ranks check that no two
        # share one (a launcher that leaves LOCAL_RANK unset, or a HIP_VISIBLE_DEVICES mask, would silently stack the ranks on cuda:0
        # and the "N-GPU" number would be an N-process number).  On failure EVERY rank says what it sees and the job exits non-zero.
        pr = torch.cuda.get_device_properties(local)
        import socket
        mine = {"rank": rank, "local_rank": local, "host": socket.gethostname(), "current_device": torch.cuda.current_device(),
                "pci": (getattr(pr, "pci_domain_id", None), getattr(pr, "pci_bus_id", None), getattr(pr, "pci_device_id", None)),
                "uuid": str(getattr(pr, "uuid", "")), "visible": torch.cuda.device_count(), "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend()}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        # two ranks share a GPU only if EVERYTHING they report about it is equal (device index in the process's visible set, PCI address,
        # uuid): attributes a driver build leaves empty or equal for all devices can then never turn N good ranks into a false alarm
        ids = [(d["host"], d["current_device"], d["visible"], tuple(d["pci"]), d["uuid"]) for d in seen]      # (only ranks of one host can collide)
        if len(set(ids)) != world or mine["rccl_ranks"] != world:
            print(f"[bench] rank {rank}: ranks do not sit on {world} distinct GPUs -- this rank: {mine}; all: {ids}", file=sys.stderr, flush=True)
            sys.exit(3)

    cls, kw, gflop_fwd, feats = MODELS[a.model]
    torch.manual_seed(0)                                  # identical weights on every rank
    model = getattr(A, cls)(scale_factor=a.scale, precision=PREC[a.dtype], **kw).to(dev)
    batch = T.synthetic_batch(a.batch, 3, a.patch, a.scale, 1234 + rank, dev)
    params = [p for p in model.parameters() if p.requires_grad]
    ddp = (world > 1 or force_ddp) and not a.inference
    use_torch_ddp = ddp and os.environ.get("SRK_USE_TORCH_DDP") == "1"
    # hipGraph modes.  N = 1: the whole step is ONE graph.  N > 1 (default, "segmented"): forward + loss + backward +
    # gradient packing are one graph, the bucketed all-reduce (RCCL over xGMI) and the optimizer step (three launches) are issued
    # eagerly behind it -- no collective is ever inside a capture, so the N > 1 run keeps the graph's launch rate without
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

    # fp16 (BASELINE config 5; the reference's `precision: 16` = autocast + GradScaler): dynamic loss scaling with its state on the
    # device (optim.DeviceGradScaler), so the fp16 step is a hipGraph like the bf16 one; without scaling the L1 gradient
    # (1 / 28M per element at batch 256) is below fp16's smallest subnormal
    scaler = A.optim.DeviceGradScaler(dev) if (a.dtype == "f16" and isinstance(opt, A.optim.Adam)) else None

    def opt_step():
        if scaler is None:
            opt.step()
        else:
            opt.step(grad_scaler=scaler)

    def fwd_bwd():
        opt.zero_grad(set_to_none=True)
        sr = net(batch["lr"])
        loss = model._calculate_losses(img_sr=sr, img_hr=batch["hr"])["loss"]
        A.ops.backward(loss if scaler is None else scaler.scale(loss))
        return loss

    def train_step():
        loss = fwd_bwd()
        if gsync is not None:
            gsync.sync()
        opt_step()
        return loss

    def infer_step():
        with torch.no_grad():
            return net(batch["lr"])

    step = infer_step if a.inference else train_step
    last = {}
    graphs = None
    ogs = None
    nseg = T.auto_segments(model) if (ddp and not use_torch_ddp and mode == "segmented") else 1
    if nseg > 1:
        # large models on several ranks: the backward pass as `nseg` graph segments with the bucket all-reduces between them
        # (trainer.OverlappedGraphStep); SRK_DDP_SEGMENTS=1 keeps the single backward graph
        try:
            with torch.cuda.stream(side):
                for _ in range(8):
                    step()
                gsync.detach()
                ogs = T.OverlappedGraphStep(model, opt, nseg, scaler=scaler)
                ogs.prepare(batch)
                for _ in range(2):
                    ogs.eager_step(batch)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            ogs.capture(batch)
            last["out"] = ogs.loss
            graphs, mode = tuple(ogs.graphs), f"segmented_overlap{ogs.nseg}"
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] segmented-overlap capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graphs, mode = None, "eager"
            step = (lambda: ogs.eager_step(batch)) if (ogs is not None and ogs.gsync is not None) else step
            torch.cuda.synchronize()
    elif mode != "eager":
        try:
            with torch.cuda.stream(side):
                for _ in range(11 if ddp else 3):
                    step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            cem = "thread_local" if ddp else "global"     # RCCL's watchdog thread polls events while this thread captures
            if mode == "full":
                g = torch.cuda.CUDAGraph()
                with A.ops.graph_capture(g, stream=side, capture_error_mode=cem):
                    last["out"] = step()
                graphs = (g,)
            else:
                # ONE graph per step also with several ranks: [optimizer step on the gradients the previous step reduced] + forward +
                # backward + packing; only the bucket all-reduce is issued eagerly between two replays.  (Round 3 issued the optimizer
                # step's three launches eagerly behind the all-reduce: 15 us of host-paced launches per step, half of the multi-rank
                # structure's 3 % overhead at batch 16.)  Same arithmetic, shifted: step k's update opens replay k + 1; one eager
                # forward + backward + reduce primes the first replay, one eager optimizer step closes the timed region's last.
                with torch.cuda.stream(side):
                    fwd_bwd()
                    gsync.sync()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                if hasattr(opt, "reserve_capture_tables"):
                    opt.reserve_capture_tables()
                ga = torch.cuda.CUDAGraph()
                with A.ops.graph_capture(ga, stream=side, capture_error_mode=cem):
                    opt_step()
                    last["out"] = fwd_bwd()
                    gsync.pack()
                graphs = (ga, "opt_first")
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"[bench] hipGraph capture ({mode}) failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
            graphs, mode = None, "eager"
            torch.cuda.synchronize()

    def run_one():
        if graphs is None:
            last["out"] = step()
        elif ogs is not None:
            ogs.step()
        elif len(graphs) == 1:
            graphs[0].replay()
        else:
            graphs[0].replay()              # [update from the previous step's reduced gradients] forward, backward, packing
            gsync.reduce()

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

    # the collective alone (all buckets back to back, every rank), HIP events on the launch stream: what a step exposes when nothing
    # overlaps it -- so that the first multi-GPU run explains itself (`config.allreduce_ms`, `config.rccl_ranks`)
    ar_ms, ar_sync = None, (ogs.gsync if (ogs is not None and ogs.gsync is not None) else gsync)
    if ddp and ar_sync is not None and ar_sync.flat is not None:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            ar_sync.reduce()
        e0.record()
        for _ in range(10):
            ar_sync.reduce()
        e1.record()
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 10.0

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
        # what the path executes against what the reference's module graph counts (executed_gflop_fwd): the last upsampler stage + tail
        # conv of EDSR / RCAN / RDN run as one 5x5 convolution on the 16-bit path
        gx = executed_gflop_fwd(a.model, gflop_fwd, a.patch, a.scale, a.dtype, A.ops._HR_COLLAPSE) * (1.0 if a.inference else 3.0)
        out["model_mfma_frac_executed"] = round(value / world * gx / 1e3 / PEAK_TFLOPS[a.dtype], 4)
        out["config"]["gflop_per_patch"] = {"reference_graph": round(flop_per_patch, 3), "executed": round(gx, 3)}
        out["config"]["hr_stage"] = ("collapsed: conv3x3(64->256) + PixelShuffle(2) + conv3x3(64->3) as one 5x5 conv 64->12 with exact border terms "
                                     "(ops.hr_tail; SRK_NO_HR_COLLAPSE=1 = layer by layer)"
                                     if (gx != flop_per_patch) else "layer by layer")
        if scaler is not None:
            out["config"]["loss_scale"] = {"kind": "dynamic, device-resident (optim.DeviceGradScaler)", "scale": scaler.get_scale(), "skipped_steps": scaler.skipped_steps}
        if ar_ms is not None:
            out["config"]["allreduce_ms"] = round(ar_ms, 4)
            out["config"]["allreduce_buckets"] = [int((b[1] - b[0]) * 4) for b in ar_sync.buckets]
        if ddp:
            out["config"]["rccl_ranks"] = dist.get_world_size() if dist.is_initialized() else 1
            out["config"]["backend"] = dist.get_backend() if dist.is_initialized() else None
        if sustained is not None:
            out["sustained_ms_per_step"] = round(sustained * 1e3, 4)
            out["sustained_value"] = round(a.batch * world / sustained, 2)
        if not a.no_roofline:
            try:
                # (several ranks: rank 0 alone must not run a backward pass through a model whose gradient hooks / flat buffer belong to the
                # process group -- a hook-launched all-reduce on one rank would wait for the others forever; the N = 1 line carries `in_step`)
                ins = None if (a.inference or ddp) else in_step_body(A, model, a.batch, a.patch, a.dtype)
                out["roofline"] = dominant_kernel_roofline(A, a.model, a.batch, a.patch, feats, a.dtype, in_step=ins)
            except Exception as e:  # noqa: BLE001
                out["roofline"] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and not a.no_other_configs and not a.inference and a.model == "edsr_baseline" and a.batch == 256:
            # BASELINE.json configs 2-5 (+ SURVEY 8(f)'s SRResNet / D-DBPN) at the reference's batch (16 patches per GPU), a few seconds each, after the timed region
            oc = []
            graphs = None                      # (frees the default line's graph and its private memory pool)
            torch.cuda.empty_cache()
            for name in ("edsr_baseline", "rcan", "edsr_large", "wdsr_b", "rdn_b", "srresnet", "ddbpn"):
                try:
                    e_, m_ = quick_train_rate(A, T, name, 16, a.patch, a.scale, a.dtype, keep_model=True)
                    try:      # the dominant kernel of THAT model at THAT batch against its roofline (short sustained window), in the step where the model has a trunk
                        ins_ = in_step_body(A, m_, 16, a.patch, a.dtype, sustain_s=0.25)
                        del m_
                        r_ = dominant_kernel_roofline(A, name, 16, a.patch, MODELS[name][3], a.dtype, iters=60, sustain_s=0.25, in_step=ins_)
                        e_["roofline"] = {k: r_[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "us_per_launch", "where", "traffic", "traffic_kernel",
                                                                      "algorithmic_bytes_per_launch", "algorithmic_bytes_are", "traffic_over_algorithmic") if k in r_}
                        if ins_ is not None:
                            e_["roofline"]["in_step"] = {k: ins_[k] for k in ("convs", "convs_per_launch", "fwd_us_per_conv", "dgrad_us_per_conv", "wgrad_us_per_layer")}
                        iso_ = r_.get("isolated") or {}
                        e_["roofline"]["isolated"] = {k: iso_[k] for k in ("frac", "variants_us", "step_weighted_frac") if k in iso_}
                    except Exception as e:  # noqa: BLE001
                        e_["roofline"] = {"error": f"{type(e).__name__}: {e}"}
                    oc.append(e_)
                except Exception as e:  # noqa: BLE001
                    oc.append({"model": name, "batch": 16, "error": f"{type(e).__name__}: {e}"})
                torch.cuda.empty_cache()
            if a.dtype == "bf16":
                # BASELINE.json config 5 is fp16 (configs/all.yml:122): WDSR-B and RDN-B at the reference's batch in fp16
                for name in ("wdsr_b", "rdn_b"):
                    try:
                        e_ = quick_train_rate(A, T, name, 16, a.patch, a.scale, "f16")
                        e_["dtype"] = "f16"
                        oc.append(e_)
                    except Exception as e:  # noqa: BLE001
                        oc.append({"model": name, "batch": 16, "dtype": "f16", "error": f"{type(e).__name__}: {e}"})
                    torch.cuda.empty_cache()
            out["other_configs"] = oc
            if A.ops._HR_COLLAPSE and gx != flop_per_patch:
                # the same step with the HR stage layer by layer (every multiply-add of the reference's module graph executed), for comparison
                A.ops._HR_COLLAPSE = False
                try:
                    lw = quick_train_rate(A, T, a.model, a.batch, a.patch, a.scale, a.dtype, seconds=1.0)
                    out["layerwise_hr_stage"] = {"value": lw["value"], "unit": lw["unit"], "ms_per_step": lw["ms_per_step"], "model_mfma_frac": lw["model_mfma_frac"], "model_mfma_frac_executed": lw.get("model_mfma_frac_executed"),
                                                 "note": "same model, batch and step with SRK_NO_HR_COLLAPSE=1: the upsampler's last stage and the tail conv as two layers"}
                except Exception as e:  # noqa: BLE001
                    out["layerwise_hr_stage"] = {"error": f"{type(e).__name__}: {e}"}
                finally:
                    A.ops._HR_COLLAPSE = True
                torch.cuda.empty_cache()
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
