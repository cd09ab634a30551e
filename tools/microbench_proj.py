#!/usr/bin/env python3
"""Times D-DBPN's direct projection kernels (csrc/proj.hip): srk_proj_up, srk_proj_down, srk_proj_wgrad at N x 48 x 48 LR pixels,
against the HBM time of their algorithmic bytes (the 32-channel HR tensor + the LR tensor, once).
usage: microbench_proj.py [--n 16] [--hw 48] [--iters 20] [--only up|down|wgrad] [--prelu]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=16); p.add_argument("--hw", type=int, default=48)
p.add_argument("--iters", type=int, default=20); p.add_argument("--only", default=""); p.add_argument("--prelu", action="store_true")
a = p.parse_args()
ops, L = A.ops, A._lib
lib = L.load()
dt, dev = torch.bfloat16, torch.device("cuda")
n, hw = a.n, a.hw
xl = (torch.rand(n, hw, hw, 32, device=dev) - 0.5).to(dt)
xh = (torch.rand(n, 4 * hw, 4 * hw, 32, device=dev) - 0.5).to(dt)
w4 = (torch.rand(32, 32, 8, 8, device=dev) - 0.5) * 0.05
b32 = torch.zeros(32, device=dev)
sl = torch.full((32,), 0.25, device=dev) if a.prelu else None
half = lib.srk_proj_pack_bytes() // 2
wpk = torch.empty(2 * half, dtype=torch.uint8, device=dev)
L.check(lib.srk_proj_pack(w4.data_ptr(), wpk.data_ptr(), ops._DT[dt], torch.cuda.current_stream().cuda_stream), "srk_proj_pack")
dw, db = torch.empty(32, 32, 8, 8, device=dev), torch.empty(32, device=dev)
scratch = torch.empty(lib.srk_proj_wgrad_scratch_floats(n, hw, hw), dtype=torch.float32, device=dev)


def wg():
    L.call("srk_proj_wgrad", L.ProjWgradArgs(xh=xh.data_ptr(), xh_pitch=32, g=xl.data_ptr(), g_pitch=32, scratch=scratch.data_ptr(), dw=dw.data_ptr(),
                                             accumulate=0, N=n, H=hw, W=hw, dtype=ops._DT[dt], db=db.data_ptr(), bias_side=2, db_accumulate=0),
           torch.cuda.current_stream().cuda_stream)


fns = {"up": lambda: ops._proj_launch(xl, wpk[half:], b32, True, sl, a.prelu), "down": lambda: ops._proj_launch(xh, wpk[:half], b32, False, sl, a.prelu),
       "wgrad": wg}
px = n * hw * hw
by = 17.0 * px * 32 * 2
for name, fn in fns.items():
    if a.only and a.only != name:
        continue
    for _ in range(3): fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st): fn()
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(a.iters): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * a.iters)
    extra = (16.0 * px * 32 * 2 if (a.prelu and name == "up") else (px * 32 * 2.0 if (a.prelu and name == "down") else 0.0))
    print(f"proj_{name:5s} n={n} {hw}x{hw}{' +prelu' if a.prelu and name != 'wgrad' else ''}: {us:7.2f} us/launch  "
          f"{(by + extra) / us / 1e3:7.1f} GB/s of algorithmic bytes ({(by + extra) / us / 1e3 / 8000:.3f} of 8 TB/s), {2.0 * px * 2048 * 32 / us / 1e6:6.1f} TFLOP/s")
