#!/usr/bin/env python3
"""Isolated launches of ONE flavour of the dominant kernel at the bench shape (for rocprofv3 --kernel-trace):
plain = conv+bias+ReLU, residual = conv*0.1 + residual, mask = data gradient with ReLU mask, wgrad = 8 queued weight
gradients per flush (grouped launch + grouped finalize).  `--seconds S`: keep replaying for S seconds (the sustained state bench.py's
`variants_us` is quoted on).  usage: microbench_variants.py --n 256 --variant plain --iters 30 --seconds 1.5"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=256); p.add_argument("--variant", default="plain"); p.add_argument("--iters", type=int, default=30)
p.add_argument("--seconds", type=float, default=0.0)
a = p.parse_args()
dev, dt, F = torch.device("cuda"), torch.bfloat16, 64
x = (torch.rand(a.n, 48, 48, F, device=dev) - 0.5).to(dt)
x2 = (torch.rand(a.n, 48, 48, F, device=dev) - 0.5).to(dt)
w = torch.nn.Parameter((torch.rand(F, F, 3, 3, device=dev) - 0.5) * 0.05); b = torch.nn.Parameter(torch.zeros(F, device=dev))
pk, pkd = A.ops.pack_conv(w, b, dt), A.ops.pack_conv(w, None, dt, dgrad=True)
out = torch.empty_like(x)
kw = dict(N=a.n, H=48, W=48, Cin=F, Cout=F, out=out)
ws = [torch.nn.Parameter(torch.zeros(F, F, 3, 3, device=dev)) for _ in range(8)]
bs = [torch.nn.Parameter(torch.zeros(F, device=dev)) for _ in range(8)]
def wg():
    with A.ops.hold_wgrads():
        for wi, bi in zip(ws, bs):
            A.ops.wgrad(x, x2, wparam=wi, bparam=bi, N=a.n, H=48, W=48, Cin=F, Cout=F, k=3, w_shape=(F, F, 3, 3))
# what a training step issues (ops.ConvChainFn): conv + ReLU also writes the ReLU sign bits, the data gradient behind it masks with them
A.ops.conv_raw(x, pk, relu=True, relu_bits="want", **kw)
bits = out.__dict__.pop("_srk_bits", None)
def plain():
    A.ops.conv_raw(x, pk, relu=True, relu_bits="want", **kw); out.__dict__.pop("_srk_bits", None)
fn = {"plain": plain, "plain_nobits": lambda: A.ops.conv_raw(x, pk, relu=True, **kw), "residual": lambda: A.ops.conv_raw(x, pk, scale=0.1, res=x2, **kw),
      "mask": lambda: A.ops.conv_raw(x, pkd, mask=x2, mask_bits=bits, scale=0.1, use_bias=False, **kw),
      "mask_activation": lambda: A.ops.conv_raw(x, pkd, mask=x2, scale=0.1, use_bias=False, **kw), "wgrad": wg}[a.variant]
for _ in range(3): fn()
torch.cuda.synchronize()
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st): fn()
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(a.iters): fn()
for _ in range(3): g.replay()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
while time.perf_counter() - t0 < a.seconds:
    for _ in range(4): g.replay()
    torch.cuda.synchronize()
print(a.variant, "done")
