#!/usr/bin/env python3
"""Channel-attention kernels (srk_ca_pool / srk_ca_apply / srk_ca_bwd_apply) against the HBM roofline.
usage: microbench_ca.py [--n 64] [--hw 48] [--c 64]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
from sr_amd import _lib as L
p = argparse.ArgumentParser(); p.add_argument("--n", type=int, default=64); p.add_argument("--hw", type=int, default=48)
p.add_argument("--c", type=int, default=64); p.add_argument("--iters", type=int, default=50)
a = p.parse_args()
dev = torch.device("cuda"); dt = torch.bfloat16; DT = 0
n, hw, c, cr = a.n, a.hw * a.hw, a.c, a.c // 16
t = (torch.rand(n, a.hw, a.hw, c, device=dev) - 0.5).to(dt); x = (torch.rand_like(t.float()) - 0.5).to(dt); g = (torch.rand_like(t.float()) - 0.5).to(dt)
out = torch.empty_like(t); gt = torch.empty_like(t)
sums = torch.zeros(n, c, device=dev); gsum = torch.zeros(n, c, device=dev); s = torch.empty(n, c, device=dev); z = torch.empty(n, cr, device=dev)
w1 = torch.rand(cr, c, device=dev) * 0.1; b1 = torch.zeros(cr, device=dev); w2 = torch.rand(c, cr, device=dev) * 0.1; b2 = torch.zeros(c, device=dev)
dw1 = torch.zeros(cr * c, device=dev); db1 = torch.zeros(cr, device=dev); dw2 = torch.zeros(c * cr, device=dev); db2 = torch.zeros(c, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream
pool = lambda: L.call("srk_ca_pool", L.CaPoolArgs(t=t.data_ptr(), t_pitch=c, t_coff=0, u=0, u_pitch=0, u_coff=0, sums=sums.data_ptr(), N=n, HW=hw, C=c, dtype=DT), st())
pool2 = lambda: L.call("srk_ca_pool", L.CaPoolArgs(t=t.data_ptr(), t_pitch=c, t_coff=0, u=g.data_ptr(), u_pitch=c, u_coff=0, sums=gsum.data_ptr(), N=n, HW=hw, C=c, dtype=DT), st())
apply_ = lambda: L.call("srk_ca_apply", L.CaApplyArgs(t=t.data_ptr(), t_pitch=c, t_coff=0, res=x.data_ptr(), res_pitch=c, res_coff=0, sums=sums.data_ptr(),
    w1=w1.data_ptr(), b1=b1.data_ptr(), w2=w2.data_ptr(), b2=b2.data_ptr(), s_out=s.data_ptr(), z_out=z.data_ptr(), out=out.data_ptr(), out_pitch=c, out_coff=0,
    N=n, HW=hw, C=c, Cr=cr, dtype=DT), st())
bwd = lambda: L.call("srk_ca_bwd_apply", L.CaBwdArgs(g=g.data_ptr(), g_pitch=c, g_coff=0, gsum=gsum.data_ptr(), sums=sums.data_ptr(), s=s.data_ptr(), z=z.data_ptr(),
    w1=w1.data_ptr(), w2=w2.data_ptr(), dw1=dw1.data_ptr(), db1=db1.data_ptr(), dw2=dw2.data_ptr(), db2=db2.data_ptr(), gt=gt.data_ptr(), gt_pitch=c, gt_coff=0,
    N=n, HW=hw, C=c, Cr=cr, dtype=DT), st())
elems = n * hw * c * 2
for name, f, nbytes in (("ca_pool (1 read)", pool, elems), ("ca_pool t*g (2 reads)", pool2, 2 * elems), ("ca_apply (2 reads + 1 write)", apply_, 3 * elems), ("ca_bwd_apply (1 read + 1 write)", bwd, 2 * elems)):
    for _ in range(3): f()
    torch.cuda.synchronize()
    sd = torch.cuda.Stream(); sd.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(sd): f()
    torch.cuda.current_stream().wait_stream(sd); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(a.iters): f()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / a.iters
    print(f"{name:34s} n={n} {a.hw}x{a.hw}x{c}: {us:8.2f} us  {nbytes/us/1e3:8.1f} GB/s algorithmic ({nbytes/1e6:.1f} MB)")
