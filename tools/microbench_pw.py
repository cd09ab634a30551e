#!/usr/bin/env python3
"""Times the fused WDSR-B pointwise kernels (csrc/pw_chain.hip) on a flat pixel list: srk_pw_forward, srk_pw_backward,
srk_pw_wgrad, against their MFMA-bound time.  usage: microbench_pw.py [--n 256] [--f 128] [--iters 20] [--only fwd|bwd|wgrad]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=256); p.add_argument("--f", type=int, default=128)
p.add_argument("--iters", type=int, default=20); p.add_argument("--only", default="")
a = p.parse_args()
ops = A.ops
dt, dev = torch.bfloat16, torch.device("cuda")
f, chid, cmid = a.f, 6 * a.f, int(0.8 * a.f)
cz = ops.pad16(cmid)
x = (torch.rand(a.n, 48, 48, f, device=dev) - 0.5).to(dt)
gz = (torch.rand(a.n, 48, 48, cz, device=dev) - 0.5).to(dt)
g = (torch.rand(a.n, 48, 48, f, device=dev) - 0.5).to(dt)
w1 = torch.nn.Parameter((torch.rand(chid, f, 1, 1, device=dev) - 0.5) * 0.2); b1 = torch.zeros(chid, device=dev)
w2 = torch.nn.Parameter((torch.rand(cmid, chid, 1, 1, device=dev) - 0.5) * 0.1); b2 = torch.zeros(cmid, device=dev)
pk = ops.pw_pack(w1, b1, w2, b2, dt)
z = torch.empty_like(gz); gx = torch.empty_like(x)
P = a.n * 48 * 48
rp = 128 if f == 128 else 64          # padded rows of conv 2
fl = {"fwd": 2.0 * P * chid * (f + rp), "bwd": 2.0 * P * chid * (f + rp + f), "wgrad": 2.0 * P * chid * (f + rp + rp + f)}
fns = {"fwd": lambda: ops.pw_forward_raw(x, pk, z), "bwd": lambda: ops.pw_backward_raw(x, gz, pk, gx, res=g),
       "wgrad": lambda: ops.pw_wgrad_raw(x, gz, pk, tuple(w1.shape), tuple(w2.shape))}
for name, fn in fns.items():
    if a.only and a.only != name:
        continue
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st): fn()
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(a.iters): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"pw_{name:5s} n={a.n} f={f}: {us:8.2f} us  {fl[name] / us / 1e6:7.1f} TFLOP/s issued ({fl[name] / us / 1e6 / 2500:.3f} of 2.5 PF; MFMA work incl. padding / recompute)")
