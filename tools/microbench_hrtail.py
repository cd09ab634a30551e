#!/usr/bin/env python3
"""The collapsed HR stage alone (ops.hr_tail: csrc/hr_tail.hip + the 5x5 kernels of csrc/conv_lk.hip) at the bench shape -- EDSR-baseline's
last upsampler stage + tail conv: 64 channels @96x96 -> 3 channels @192x192 -- forward + backward, for rocprofv3 runs (kernel trace / PMC).
usage: microbench_hrtail.py --n 256 --iters 10 [--layerwise]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=256); p.add_argument("--iters", type=int, default=10); p.add_argument("--hw", type=int, default=96)
p.add_argument("--layerwise", action="store_true", help="the two layers one after the other (conv + PixelShuffle store, tail conv)")
a = p.parse_args()
dev, dt = torch.device("cuda"), torch.bfloat16
torch.manual_seed(0)
x = (torch.rand(a.n, a.hw, a.hw, 64, device=dev) - 0.5).to(dt).requires_grad_(True)
up = torch.nn.Conv2d(64, 256, 3, padding=1).to(dev)
tail = torch.nn.Conv2d(64, 3, 3, padding=1).to(dev)
g = torch.rand(a.n, 3, 2 * a.hw, 2 * a.hw, device=dev) - 0.5
post = torch.tensor([0.4488, 0.4371, 0.4040], device=dev)


def step():
    for q in (up.weight, up.bias, tail.weight, tail.bias, x):
        q.grad = None
    if a.layerwise:
        u = A.ops.conv(x, up.weight, up.bias, ps_r=2)
        y = A.ops.tail_conv(u, tail.weight, tail.bias, post_add=post)
    else:
        y = A.ops.hr_tail(x, up.weight, up.bias, tail.weight, tail.bias, post_add=post)
    y.backward(g)


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    step()
torch.cuda.synchronize()
print(f"hr stage {'layer by layer' if a.layerwise else 'collapsed'} n={a.n} {a.hw}x{a.hw}: {(time.perf_counter() - t0) / a.iters * 1e6:.1f} us per forward + backward (eager launches)")
