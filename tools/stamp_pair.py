#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of workgroup 0 of conv_pair_kernel (compute wave 0 and DMA-only wave 4).
Needs the stamp build: `make -C sr-pytorch-lightning_amd/csrc stamp`."""
import os, sys
os.environ.setdefault("SRK_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libsrk_stamp.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda"); dt = torch.bfloat16
x = (torch.rand(n, 48, 48, 64, device=dev) - 0.5).to(dt)
ws = [torch.nn.Parameter((torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.05) for _ in range(2)]
pk = [A.ops.pack_conv(w, None, dt) for w in ws]
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
class P: pass
p2 = P(); p2.wpk = pk[1].wpk; p2.bias = stamps.view(torch.float32)
mid, o = torch.empty_like(x), torch.empty_like(x)
def run():
    A.ops.conv_pair_raw(x, pk[0], p2, out=o, relu_mid=True, mid=mid, scale_out=0.1, res=x)
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1)*1e3/20:.2f} us per launch")
st = stamps.cpu().numpy()
names = ["entry", "x tile issued", "slabs issued", "mask requested", "own DMA landed", "barrier", "conv1 done", "mid epilogue done",
         "drain barrier", "b0 barrier", "conv2 done", "stores issued", "stores done"]
for w in (0, 1):
    t = st[w * 16:(w + 1) * 16]
    print(("compute wave 0: " if w == 0 else "DMA wave 4:     ") + "; ".join(f"{nm} {t[i]-t[0]}" for i, nm in enumerate(names)))
