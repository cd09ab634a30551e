#!/usr/bin/env python3
"""Dominant conv with the SAME buffers every launch (Infinity-Cache resident) vs rotating over K buffer pairs
(working set K x 151 MB at n=256: HBM resident).  Tells how memory-bound the kernel is."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda"); dt = torch.bfloat16
w = torch.nn.Parameter((torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.05); b = torch.nn.Parameter(torch.zeros(64, device=dev))
pk = A.ops.pack_conv(w, b, dt)
for K in (1, 2, 4, 8):
    xs = [(torch.rand(n, 48, 48, 64, device=dev) - 0.5).to(dt) for _ in range(K)]
    outs = [torch.empty_like(xs[0]) for _ in range(K)]
    f = lambda i: A.ops.conv_raw(xs[i % K], pk, N=n, H=48, W=48, Cin=64, Cout=64, out=outs[i % K], relu=True)
    for i in range(K): f(i)
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st): f(0)
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    iters = 48
    with torch.cuda.graph(g):
        for i in range(iters): f(i)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    fl = 2.0 * n * 48 * 48 * 64 * 64 * 9
    print(f"n={n} buffers={K} (working set {K * 2 * n * 2304 * 128 / 1e6:.0f} MB): {us:.2f} us/launch  {fl / us / 1e6:.0f} TFLOP/s  {2 * n * 2304 * 128 / us / 1e3:.0f} GB/s algorithmic")
