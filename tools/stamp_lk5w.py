#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of wave 0 of workgroup 0 of lk5_wgrad_kernel per tile (stamp build: make -C .../csrc stamp)."""
import os, sys, ctypes as C
os.environ.setdefault("SRK_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libsrk_stamp.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
from sr_amd import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev, dt = torch.device("cuda"), torch.bfloat16
x = (torch.rand(n, 96, 96, 64, device=dev) - 0.5).to(dt)
g = (torch.rand(n, 96, 96, 16, device=dev) - 0.5).to(dt)
a = L.WgradArgs(x=x.data_ptr(), x_pitch=64, x_coff=0, dy=g.data_ptr(), dy_pitch=16, dy_coff=0, N=n, H=96, W=96, Cin=64, Cout=16, KH=5, KW=5,
                dwp=0, dbp=0, nslabs=0, dtype=0)
lib = L.load()
ns = lib.srk_wgrad_slabs(a)
per = 25 * 64 * 16
scratch = torch.zeros(ns * per + ns * 16 + 4096, dtype=torch.float32, device=dev)
a.dwp, a.dbp, a.nslabs = scratch.data_ptr(), scratch.data_ptr() + ns * per * 4, ns
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    L.call("srk_conv2d_wgrad", a, st)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); L.call("srk_conv2d_wgrad", a, st); e1.record(); torch.cuda.synchronize()
raw = scratch[ns * per:].view(torch.int64)[ns * 8:ns * 8 + 24 * 8].cpu().numpy().reshape(24, 8)
print(f"launch {e0.elapsed_time(e1) * 1e3:.1f} us, {ns} slabs; s_memtime ticks relative to tile 0's first stamp")
t0 = raw[0, 0]
print("tile   start  waited barrier dma-issued mfma-done | tile length")
for s in range(24):
    if raw[s, 0] == 0: break
    r = raw[s] - t0
    nxt = (raw[s + 1, 0] - raw[s, 0]) if s + 1 < 24 and raw[s + 1, 0] else 0
    print(f"{s:4d} {r[0]:7d} {r[1]:7d} {r[2]:7d} {r[3]:10d} {r[4]:9d} | {nxt}")
