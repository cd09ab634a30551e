#!/usr/bin/env python3
"""Isolated launches of one conv configuration (for rocprofv3 --pmc / timing).
usage: microbench_conv.py [--n 64] [--hw 48] [--cin 64] [--cout 64] [--k 3] [--iters 20] [--mode fwd|wgrad] [--dtype bf16]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=64); p.add_argument("--hw", type=int, default=48)
p.add_argument("--cin", type=int, default=64); p.add_argument("--cout", type=int, default=64)
p.add_argument("--k", type=int, default=3); p.add_argument("--iters", type=int, default=20)
p.add_argument("--mode", default="fwd"); p.add_argument("--dtype", default="bf16")
p.add_argument("--relu", type=int, default=1); p.add_argument("--res", type=int, default=0)
a = p.parse_args()
dt = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}[a.dtype]
dev = torch.device("cuda")
x = (torch.rand(a.n, a.hw, a.hw, a.cin, device=dev) - 0.5).to(dt)
w = torch.nn.Parameter((torch.rand(a.cout, a.cin, a.k, a.k, device=dev) - 0.5) * 0.05)
b = torch.nn.Parameter(torch.zeros(a.cout, device=dev))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if a.mode == "fwd":
    pk = A.ops.pack_conv(w, b, dt)
    out = torch.empty(a.n, a.hw, a.hw, A.ops.pad16(a.cout), device=dev, dtype=dt)
    res = torch.zeros_like(out) if a.res else None
    f = lambda: A.ops.conv_raw(x, pk, N=a.n, H=a.hw, W=a.hw, Cin=a.cin, Cout=out.shape[3], out=out, relu=bool(a.relu), res=res)
else:
    dy = (torch.rand(a.n, a.hw, a.hw, A.ops.pad16(a.cout), device=dev) - 0.5).to(dt)
    f = lambda: A.ops.wgrad_raw(x, dy, N=a.n, H=a.hw, W=a.hw, Cin=a.cin, Cout=dy.shape[3], k=a.k, w_shape=tuple(w.shape))
for _ in range(3): f()
torch.cuda.synchronize()
# replay the launches from ONE hipGraph so that the Python/ctypes launch rate (~10 us) does not bound the timing
st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st):
    f()
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(a.iters): f()
g.replay(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / a.iters
fl = 2.0 * a.n * a.hw * a.hw * a.cin * a.cout * a.k * a.k
print(f"{a.mode} {a.dtype} n={a.n} {a.hw}x{a.hw} {a.cin}->{a.cout} k{a.k}: {us:.2f} us/iter  {fl/us/1e6:.1f} TFLOP/s  ws={'off' if (os.environ.get('SRK_NO_WS') and os.environ.get('SRK_DEBUG') == '1') else 'on'}")
