#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of workgroup 0 (wave 0: compute, wave 4: DMA) of conv_ks_kernel (csrc/conv_ks.hip).  Needs the stamp
build: `make -C sr-pytorch-lightning_amd/csrc stamp`.  usage: stamp_ks.py [--n 256] [--cin 112] [--cout 128] [--res 0|1]"""
import argparse, ctypes as C, os, sys
os.environ.setdefault("SRK_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libsrk_stamp.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=256); p.add_argument("--cin", type=int, default=112); p.add_argument("--cout", type=int, default=128)
p.add_argument("--res", type=int, default=0)
a = p.parse_args()
dt, dev = torch.bfloat16, torch.device("cuda")
x = (torch.rand(a.n, 48, 48, a.cin, device=dev) - 0.5).to(dt)
w = torch.nn.Parameter((torch.rand(a.cout, a.cin, 3, 3, device=dev) - 0.5) * 0.05); b = torch.nn.Parameter(torch.zeros(a.cout, device=dev))
pk = A.ops.pack_conv(w, b, dt)
out = torch.empty(a.n, 48, 48, A.ops.pad16(a.cout), device=dev, dtype=dt)
res = torch.zeros_like(out) if a.res else None
f = lambda: A.ops.conv_raw(x, pk, N=a.n, H=48, W=48, Cin=a.cin, Cout=out.shape[3], out=out, relu=not a.res, res=res)
for _ in range(3): f()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(10): f()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 10
nkb = (a.cin + 63) // 64
wgs = a.n * 9 * ((a.cout + 63) // 64)
print(f"conv_ks {a.cin} -> {a.cout} x{a.n}: {us:.1f} us per launch, {wgs} workgroups of {nkb} K-blocks ({nkb * 4608} MFMA cycles per compute wave)")
buf = (C.c_ulonglong * 64)()
assert A._lib.load().srk_ks_read_stamps(buf) == 0
t = list(buf[0:32]); t0 = t[0]
ntile = (a.n * 9 * ((a.cout + 63) // 64) + 255) // 256 if wgs > 256 else 1
kbs = [t[2 + k] - t0 for k in range(min(8, nkb * ntile)) if t[2 + k] > t0]
print(f"compute wave 0 of workgroup 0 ({ntile} tiles): first data landed + barrier @{t[1] - t0}; its first stream K-blocks end @{kbs} "
      f"(a tile's last one is followed by the tile end: epilogue to LDS, two barriers); kernel end @{t[13] - t0}")
