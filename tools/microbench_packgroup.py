#!/usr/bin/env python3
"""Time of the per-step weight-packing launch (ops.forward_scope(model._pack_group())) of a model, as a hipGraph replay.
usage: [SRK_LIB_PATH=...] python tools/microbench_packgroup.py [edsr_baseline rcan edsr_large ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import sr_amd as A
names = sys.argv[1:] or ["edsr_baseline", "rcan", "edsr_large"]
for name in names:
    cls, kw, _, _ = bench.MODELS[name]
    torch.manual_seed(0)
    m = getattr(A, cls)(scale_factor=4, precision="bf16", **kw).cuda()
    x = torch.rand(2, 3, 48, 48, device="cuda")
    for _ in range(2):                       # the group learns its entries (forward and data-gradient packs) from a step
        m(x).sum().backward()
    def pack_only():
        with A.ops.forward_scope(m._pack_group()):
            pass
    us = bench._replay_us(pack_only, 0.2)
    nparam = sum(p.numel() for p in m.parameters())
    print(f"{name}: pack launch {us:.1f} us for {nparam / 1e6:.2f} M parameters")
