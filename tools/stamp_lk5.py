#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of wave 0 of workgroup 0 of lk5_dgrad_kernel, per step of its first unit (stamp build: make -C .../csrc stamp)."""
import os, sys
os.environ.setdefault("SRK_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libsrk_stamp.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev, dt = torch.device("cuda"), torch.bfloat16
g12 = (torch.rand(n, 96, 96, 16, device=dev) - 0.5).to(dt)
weff = (torch.rand(12, 64, 5, 5, device=dev) - 0.5) * 0.05
pkd = A.ops.pack_conv(weff, None, dt, dgrad=True, cache=False)
gx = torch.empty(n, 96, 96, 64, device=dev, dtype=dt)
stamps = torch.zeros(256, dtype=torch.int64, device=dev)
kw = dict(N=n, H=96, W=96, Cin=16, Cout=64, out=gx, use_bias=False, post_add=stamps.view(torch.float32))
for _ in range(3):
    A.ops.conv_raw(g12, pkd, **kw)
torch.cuda.synchronize(); stamps.zero_(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); A.ops.conv_raw(g12, pkd, **kw); e1.record(); torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(32, 8)
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us; s_memtime ticks (~ core cycles) relative to step 0's first stamp")
t0 = st[0, 0]
print("step   start  waited barrier  mfma-loop-done | step length      (the epilogue of step s - 1 is issued inside the MFMA loop of step s)")
for s in range(24):
    if st[s, 0] == 0: break
    r = st[s] - t0
    nxt = (st[s + 1, 0] - st[s, 0]) if s + 1 < 24 and st[s + 1, 0] else 0
    print(f"{s:4d} {r[0]:7d} {r[1]:7d} {r[2]:7d} {r[3]:15d} | {nxt}")
