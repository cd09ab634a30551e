"""Times srk_conv_pair against the two srk_conv2d launches it replaces, as a chain of 32 blocks at N x 48 x 48 (the
depth of EDSR-baseline's body, forward + backward forms).  Usage: python3 tools/microbench_pair.py [N ...]"""
import sys
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sr_amd as A

dev = torch.device("cuda")
dt = torch.bfloat16
A._lib.load()
for n in [int(v) for v in sys.argv[1:]] or [16, 32, 64]:
    x = (torch.rand(n, 48, 48, 64, device=dev) - 0.5).to(dt)
    ws = [torch.nn.Parameter((torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.05) for _ in range(4)]
    bs = [torch.nn.Parameter(torch.zeros(64, device=dev)) for _ in range(4)]
    pk = [A.ops.pack_conv(w, b, dt) for w, b in zip(ws, bs)]
    mid, o1, o2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)

    def two(a, o):
        A.ops.conv_raw(a, pk[0], N=n, H=48, W=48, Cin=64, Cout=64, out=mid, relu=True)
        A.ops.conv_raw(mid, pk[1], N=n, H=48, W=48, Cin=64, Cout=64, out=o, scale=0.1, res=a)

    def pair(a, o):
        A.ops.conv_pair_raw(a, pk[0], pk[1], out=o, relu_mid=True, mid=mid, scale_out=0.1, res=a)

    for name, fn in (("two launches", two), ("pair", pair)):
        def chain():
            a, o = x, o1
            for _ in range(32):
                fn(a, o)
                a, o = o, (o2 if o is o1 else o1)
        chain()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            chain()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20 / 32
        print(f"N={n:4d} {name:13s}: {us:7.2f} us per block ({A._lib.load().srk_conv_pair_tiles(n, 48, 48)} pair tiles)")
