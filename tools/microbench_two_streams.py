#!/usr/bin/env python3
"""Does a second stream hide the ~4 us gap between dependent launches?  2N conv launches on one stream against N + N
on two streams (independent buffers), both captured into one hipGraph each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = 20
dev = torch.device("cuda"); dt = torch.bfloat16
def mk():
    x = (torch.rand(n, 48, 48, 64, device=dev) - 0.5).to(dt)
    return x, torch.empty_like(x)
w = torch.nn.Parameter((torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.05); b = torch.nn.Parameter(torch.zeros(64, device=dev))
pk = A.ops.pack_conv(w, b, dt)
(x1, o1), (x2, o2) = mk(), mk()
f = lambda x, o: A.ops.conv_raw(x, pk, N=n, H=48, W=48, Cin=64, Cout=64, out=o, relu=True)
f(x1, o1); f(x2, o2); torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def timeit(g):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=s1):
    for _ in range(N):
        f(x1, o1); f(x2, o2)
t1 = timeit(g1)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=s1):
    s2.wait_stream(s1)
    with torch.cuda.stream(s2):
        for _ in range(N): f(x2, o2)
    for _ in range(N): f(x1, o1)
    s1.wait_stream(s2)
t2 = timeit(g2)
print(f"n={n}: {2*N} launches on one stream {t1:.1f} us ({t1/(2*N):.2f} us each); {N}+{N} on two streams {t2:.1f} us ({t2/(2*N):.2f} us each)  -> {100*(t2/t1-1):+.1f} %")
