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
stamps = torch.zeros(64 + 2048, dtype=torch.int64, device=dev)
class P: pass
p2 = P(); p2.wpk = pk[1].wpk; p2.bias = stamps.view(torch.float32)
mid, o = torch.empty_like(x), torch.empty_like(x)
MODE = int(os.environ.get("STAMP_CA", "0"))
T = A._lib.load().srk_conv_pair_tiles(1, 48, 48)
ca = dict(gsum=torch.rand(n, T, 64, device=dev), sums=torch.rand(n, T, 64, device=dev), s=torch.rand(n, 64, device=dev), z=torch.rand(n, 4, device=dev),
          w1=torch.rand(4, 64, device=dev), w2=torch.rand(64, 4, device=dev), slots=torch.empty(n, 2 * 256 + 68, device=dev),
          x2=x.clone(), b1=torch.rand(4, device=dev), b2=torch.rand(64, device=dev), s_out=torch.empty(n, 64, device=dev), z_out=torch.empty(n, 4, device=dev))
xo = torch.empty_like(x)
pool = torch.empty(n, T, 64, device=dev)
def run():
    if MODE == 1:
        A.ops.conv_pair_raw(x, pk[0], p2, out=o, mask=x, mid=mid, res=x, use_bias=True, ca_bwd=ca, xo=xo, pool=pool, pool_aux=x)
    elif MODE == 2:
        A.ops.conv_pair_raw(x, pk[0], p2, out=o, relu_mid=True, mid=mid, ca_fwd=ca, xo=xo, pool=pool)
    else:
        A.ops.conv_pair_raw(x, pk[0], p2, out=o, relu_mid=True, mid=mid, scale_out=0.1, res=x)
run(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): run()
g.replay(); torch.cuda.synchronize(); stamps.zero_(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"{e0.elapsed_time(e1)*1e3/20:.2f} us per launch")
st = stamps.cpu().numpy()
names = ["entry", "a0 issued", "tile, a1, a2 issued", "mask requested", "own DMA landed", "barrier", "conv1 done", "mid epilogue done",
         "drain barrier", "b0 barrier", "conv2 done", "stores issued", "stores done"]
for w in (0, 1):
    t = st[w * 32:(w + 1) * 32]
    print(("compute wave 0: " if w == 0 else "DMA wave 4:     ") + "; ".join(f"{nm} {t[i]-t[0]}" for i, nm in enumerate(names)))
    if t[13]:
        print("      channel-attention block: " + "; ".join(f"{nm} {t[13+i]-t[0]}" for i, nm in enumerate(["MLP operands requested", "input pieces requested", "MLP operands arrived", "MLP done", "tile transformed", "all landed", "staging barrier", "pooled mean done (wave 0)", "MLP loop done (wave 0)"])))

# every workgroup's entry / exit on the 100 MHz clock, the last two of the 20 launches
import numpy as np
tot = int(st[63])
if tot and tot % 20 == 0 and tot // 20 <= 512:
    grid = tot // 20
    def reg(k):
        r = st[64 + k * 1024: 64 + k * 1024 + 2 * grid].reshape(grid, 2).astype(np.int64)
        return r[:, 0], r[:, 1]
    e1_, x1_ = reg(1); e0_, x0_ = reg(0)            # launch 18 (even) and 19 (odd)
    t0 = e0_.min(); f = lambda v: f"{(v - t0) / 100.0:.2f}"
    print(f"all {grid} workgroups, us from the first entry of launch A (the one before the last):")
    print(f"  launch A: entries {f(e0_.min())} .. {f(e0_.max())} (median {f(np.median(e0_))}); exits {f(x0_.min())} .. {f(x0_.max())} (median {f(np.median(x0_))})")
    print(f"  launch B: entries {f(e1_.min())} .. {f(e1_.max())} (median {f(np.median(e1_))}); exits {f(x1_.min())} .. {f(x1_.max())} (median {f(np.median(x1_))})")
    print(f"  last exit of A -> first entry of B: {(e1_.min() - x0_.max()) / 100.0:.2f} us; residence: median {np.median(x0_ - e0_) / 100.0:.2f} us, min {(x0_ - e0_).min() / 100.0:.2f}, max {(x0_ - e0_).max() / 100.0:.2f}")
