#!/usr/bin/env python3
"""A residual trunk's forward chain (conv + ReLU [+ sign bits] ; conv * 0.1 + x) as ONE srk_conv_trunk launch against the same layers as
srk_conv2d launches: bit-identity of every layer's output and time per convolution (hipGraph replays, HIP events).
usage: microbench_trunk.py [--n 256] [--blocks 16] [--seconds 0.5]"""
import argparse, ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A
from sr_amd import _lib as L
p = argparse.ArgumentParser()
p.add_argument("--n", type=int, default=256); p.add_argument("--blocks", type=int, default=16); p.add_argument("--seconds", type=float, default=0.5)
p.add_argument("--hw", type=int, default=48)
a = p.parse_args()
ops = A.ops
dev, dt, F, H = torch.device("cuda"), torch.bfloat16, 64, a.hw
torch.manual_seed(0)
x0 = (torch.rand(a.n, H, H, F, device=dev) - 0.5).to(dt)
ws = [torch.nn.Parameter((torch.rand(F, F, 3, 3, device=dev) - 0.5) * 0.08) for _ in range(2 * a.blocks)]
bs = [torch.nn.Parameter((torch.rand(F, device=dev) - 0.5) * 0.1) for _ in range(2 * a.blocks)]
pks = [ops.pack_conv(w, b, dt) for w, b in zip(ws, bs)]
def buffers():
    return [torch.empty_like(x0) for _ in range(2 * a.blocks)], [torch.empty((a.n * H * H, 2), dtype=torch.int32, device=dev) for _ in range(a.blocks)]
def layer_args(outs, bits):
    arr = (L.ConvArgs * (2 * a.blocks))()
    x = x0
    for b in range(a.blocks):
        for i in range(2):
            l = 2 * b + i
            src = x if i == 0 else outs[l - 1]
            pk = pks[l]
            arr[l] = L.ConvArgs(x=src.data_ptr(), x_pitch=F, x_coff=0, x_ps=0, N=a.n, H=H, W=H, Cin=F, wpk=pk.wpk.data_ptr(), bias=pk.bias.data_ptr(),
                                CoutP=pk.CoutP, Cout=F, KH=3, KW=3, relu=int(i == 0), scale=1.0 if i == 0 else 0.1, res=0 if i == 0 else x.data_ptr(),
                                res_pitch=0 if i == 0 else F, res_coff=0, mask=0, mask_pitch=0, mask_coff=0, mask_from=0, out=outs[l].data_ptr(),
                                out_pitch=F, out_coff=0, out_mode=L.OUT_NHWC, ps_r=0, post_add=0, dtype=0, cout_real=0,
                                relu_bits=bits[b].data_ptr() if i == 0 else 0, mask_bits=0)
        x = outs[2 * b + 1]
    return arr
lib = L.load()
st = torch.cuda.current_stream().cuda_stream
o1, b1 = buffers(); o2, b2 = buffers()
arr1, arr2 = layer_args(o1, b1), layer_args(o2, b2)
nl = 2 * a.blocks
print("trunk_ok", lib.srk_conv_trunk_ok(arr2, nl))
tab = torch.empty(C.sizeof(L.ConvArgs) * nl, dtype=torch.uint8, device=dev)
L.check(lib.srk_upload_small(tab.data_ptr(), C.addressof(arr2), C.sizeof(L.ConvArgs) * nl, st), "upload")
def per_layer():
    for l in range(nl): L.call("srk_conv2d", arr1[l], st)
def trunk():
    L.check(lib.srk_conv_trunk(arr2, tab.data_ptr(), nl, st), "srk_conv_trunk")
per_layer(); trunk(); torch.cuda.synchronize()
bad = [l for l in range(nl) if not torch.equal(o1[l], o2[l])] + [100 + b for b in range(a.blocks) if not torch.equal(b1[b], b2[b])]
print("bit-identical to the per-layer launches:", not bad, bad[:8], "finite", bool(torch.isfinite(o2[-1].float()).all()), float(o2[-1].float().abs().mean()))
def timed(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < a.seconds: g.replay(); n += 1
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(max(n, 4)): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / max(n, 4) / nl
st = None
def per_layer():
    s = torch.cuda.current_stream().cuda_stream
    for l in range(nl): L.call("srk_conv2d", arr1[l], s)
def trunk():
    L.check(lib.srk_conv_trunk(arr2, tab.data_ptr(), nl, torch.cuda.current_stream().cuda_stream), "srk_conv_trunk")
for r in range(2):
    print(f"per-layer launches: {timed(per_layer):.2f} us per conv    one trunk launch: {timed(trunk):.2f} us per conv")
