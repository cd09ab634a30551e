#!/usr/bin/env python3
"""VERDICT r4 item 4(a), measured: does running each ResBlock on two half-batches back to back (x_A, h_A live in the Infinity Cache between the two
launches that use them) beat the whole batch per launch?  A forward trunk of 16 ResBlocks at 256 x 48 x 48 x 64 bf16 with ONE buffer per layer (what
a training step writes: nothing is re-read from the cache by accident), as the launches a training step issues (conv + ReLU + sign bits; conv * 0.1 +
residual), captured as a hipGraph and replayed for --seconds.  whole: 32 launches of 256 images; halves: 64 launches of 128 images, per block A then B.
usage: python3 tools/microbench_halfbatch.py [--n 256] [--blocks 16] [--seconds 1.0]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=256); p.add_argument("--blocks", type=int, default=16); p.add_argument("--seconds", type=float, default=1.0)
a = p.parse_args()
dev, dt, F = torch.device("cuda"), torch.bfloat16, 64
acts = [(torch.rand(a.n, 48, 48, F, device=dev) - 0.5).to(dt)] + [torch.empty(a.n, 48, 48, F, device=dev, dtype=dt) for _ in range(2 * a.blocks)]
w = torch.nn.Parameter((torch.rand(F, F, 3, 3, device=dev) - 0.5) * 0.05); b = torch.nn.Parameter(torch.zeros(F, device=dev))
pk = A.ops.pack_conv(w, b, dt)


def block(x, h, o, n):
    kw = dict(N=n, H=48, W=48, Cin=F, Cout=F)
    A.ops.conv_raw(x, pk, relu=True, relu_bits="want", out=h, **kw)
    h.__dict__.pop("_srk_bits", None)
    A.ops.conv_raw(h, pk, scale=0.1, res=x, out=o, **kw)


def whole():
    for k in range(a.blocks):
        block(acts[2 * k], acts[2 * k + 1], acts[2 * k + 2], a.n)


def halves():
    m = a.n // 2
    for k in range(a.blocks):
        for s in (slice(0, m), slice(m, a.n)):
            block(acts[2 * k][s], acts[2 * k + 1][s], acts[2 * k + 2][s], m)


def halves_deep():
    """half A through ALL blocks, then half B (the working set between a tensor's write and its re-read is the same as in `halves`)"""
    m = a.n // 2
    for s in (slice(0, m), slice(m, a.n)):
        for k in range(a.blocks):
            block(acts[2 * k][s], acts[2 * k + 1][s], acts[2 * k + 2][s], m)


S2 = torch.cuda.Stream()


def halves_two_streams():
    """half A's trunk and half B's trunk as two branches of the graph (fork / join by events): two chains of dependent launches that can
    fill each other's launch boundaries -- a workgroup needs a whole CU's LDS, so B's workgroups take the CUs A's leave"""
    m = a.n // 2
    s1 = torch.cuda.current_stream()
    ev = torch.cuda.Event(); ev.record(s1); S2.wait_event(ev)
    with torch.cuda.stream(S2):
        for k in range(a.blocks):
            block(acts[2 * k][m:], acts[2 * k + 1][m:], acts[2 * k + 2][m:], a.n - m)
    for k in range(a.blocks):
        block(acts[2 * k][:m], acts[2 * k + 1][:m], acts[2 * k + 2][:m], m)
    ev2 = torch.cuda.Event(); ev2.record(S2); s1.wait_event(ev2)


for name, fn in (("whole batch per launch", whole), ("two half-batches per block", halves), ("half A through the trunk, then half B", halves_deep),
                 ("half A and half B as two streams of the graph", halves_two_streams)):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < a.seconds / 2:
        g.replay(); k += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(max(k, 4)):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / max(k, 4)
    print(f"{name:42s}: {us:9.1f} us per trunk forward = {us / (2 * a.blocks):6.2f} us per conv of {a.n} images")
