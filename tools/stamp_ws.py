#!/usr/bin/env python3
"""Diagnostic: per-phase s_memtime stamps of the two wave groups of workgroup 0 of conv_ws_kernel.
Needs the stamp build of the library: `make -C sr-pytorch-lightning_amd/csrc stamp` (optionally
STAMP_DEFS=-DSRK_WS_ABLATE=<bits> for the timing ablations)."""
import os, sys
os.environ.setdefault("SRK_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libsrk_stamp.so"))   # make -C .../csrc stamp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda"); dt = torch.bfloat16
HW = int(os.environ.get("STAMP_HW", "48")); COUT = int(os.environ.get("STAMP_COUT", "64"))
x = (torch.rand(n, HW, HW, 64, device=dev) - 0.5).to(dt)
w = torch.nn.Parameter((torch.rand(COUT, 64, 3, 3, device=dev) - 0.5) * 0.05); b = torch.nn.Parameter(torch.zeros(COUT, device=dev))
pk = A.ops.pack_conv(w, b, dt)
PLANAR = COUT < 16
out = torch.empty(n, COUT, HW, HW, device=dev) if PLANAR else torch.empty(n, HW, HW, COUT, device=dev, dtype=dt)
RES = torch.zeros_like(out) if os.environ.get('STAMP_RES') else None
MASK = torch.ones_like(out) if os.environ.get('STAMP_MASK') else None
KW = dict(res=RES, mask=MASK, N=n, H=HW, W=HW, Cin=64, Cout=COUT, out=out, relu=(not PLANAR) and RES is None and MASK is None, out_mode=A._lib.OUT_PLANAR if PLANAR else A._lib.OUT_NHWC)
stamps = torch.zeros(256 + 2048, dtype=torch.int64, device=dev)
NL = int(os.environ.get('STAMP_NL', '20'))
A.ops.conv_raw(x, pk, post_add=stamps.view(torch.float32), **KW)
stamps.zero_()
torch.cuda.synchronize()
CHAIN = os.environ.get('STAMP_CHAIN')       # "fwd": the trunk's forward order on its own buffers (conv+ReLU, conv*s+residual per block); STAMP_ODD=1: one more launch (the stamped pair is then residual -> plain)
if CHAIN:
    bufs = [x] + [torch.empty_like(x) for _ in range(NL + 1)]
    def chain():
        k = 0
        for i in range(NL):
            if i % 2 == 0:
                A.ops.conv_raw(bufs[i], pk, post_add=stamps.view(torch.float32), N=n, H=HW, W=HW, Cin=64, Cout=64, out=bufs[i + 1], relu=True, out_mode=A._lib.OUT_NHWC)
            else:
                A.ops.conv_raw(bufs[i], pk, post_add=stamps.view(torch.float32), N=n, H=HW, W=HW, Cin=64, Cout=64, out=bufs[i + 1], res=bufs[i - 1], scale=0.1, out_mode=A._lib.OUT_NHWC)
    if os.environ.get('STAMP_ODD'): NL += 1; bufs.append(torch.empty_like(x))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    if CHAIN: chain()
    else:
        for _ in range(NL):
            A.ops.conv_raw(x, pk, post_add=stamps.view(torch.float32), **KW)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(int(os.environ.get('STAMP_WARM', '0'))): g.replay()
torch.cuda.synchronize(); stamps.zero_(); torch.cuda.synchronize()
e0.record(); g.replay(); e1.record()
torch.cuda.synchronize()
st = stamps.cpu().numpy()
us = e0.elapsed_time(e1) * 1e3 / NL
ticks = (st[102] - st[100]) / NL
print(f"{us:.2f} us per launch (events), {ticks:.0f} s_memtime ticks per launch period -> {ticks/us:.1f} ticks/us")
print(f"last launch: entry->first phase {st[0]-st[101]} ticks, entry->exit {st[102]-st[101]} ticks")
if st[104]:
    print(f"prologue of group 0 wave 0 (ticks from entry): setup done {st[104]-st[101]}, DMA issued {st[105]-st[101]}, own pieces landed {st[106]-st[101]}, barrier passed {st[107]-st[101]}")
for g in (0, 1):
    t = st[g * 128:(g + 1) * 128]
    t0 = st[0]
    print(f"group {g}: (start,end) cycles rel. to first stamp, per phase")
    for p in range(12):
        if t[2 * p] == 0: break
        print(f"   phase {p:2d}: start {t[2*p]-t0:7d}  end {t[2*p+1]-t0:7d}  body {t[2*p+1]-t[2*p]:6d}")




t = st[0:128]
if t[30]:
    b = t[6]
    names = ["DMA issued", "compute(0) done", "store(0) issued", "compute(1) done", "store(1) issued", "final wait done"]
    print("group 0, phase 3 (epilogue, prefetch variant) sub-stamps, ticks from phase start: " + "; ".join(f"{nm} {t[30+i]-b}" for i, nm in enumerate(names)))

if t[40]:
    b = t[4]
    print(f"group 0, phase 2 (MFMA) sub-stamps, ticks from phase start: first fragments requested {t[40]-b}; K-step 1 reached {t[41]-b}; last K-step reached {t[42]-b} "
          f"({(t[42]-t[41])/ (4*34):.2f} cycles per MFMA over steps 1..34); loop left {t[43]-b}")

# every workgroup's entry / exit on the 100 MHz clock, the last two launches (stamp build only)
import numpy as np
nl_total = int(st[255])
if nl_total:
    grid = nl_total // (NL)
    if grid * NL == nl_total and grid <= 512:
        last = ((NL - 1) & 1); prev = last ^ 1
        def reg(k):
            r = st[256 + k * 1024: 256 + k * 1024 + 2 * grid].reshape(grid, 2).astype(np.int64)
            return r[:, 0], r[:, 1]
        e1_, x1_ = reg(last); e0_, x0_ = reg(prev)
        t0 = e0_.min()
        f = lambda v: f"{(v - t0) / 100.0:.2f}"
        print(f"all {grid} workgroups, us from the first entry of the launch before the last (100 MHz clock):")
        print(f"  launch A: entries {f(e0_.min())} .. {f(e0_.max())} (median {f(np.median(e0_))}); exits {f(x0_.min())} .. {f(x0_.max())} (median {f(np.median(x0_))})")
        print(f"  launch B: entries {f(e1_.min())} .. {f(e1_.max())} (median {f(np.median(e1_))}); exits {f(x1_.min())} .. {f(x1_.max())} (median {f(np.median(x1_))})")
        print(f"  last exit of A -> first entry of B: {(e1_.min() - x0_.max()) / 100.0:.2f} us; workgroup residence: median {np.median(x0_ - e0_) / 100.0:.2f} us, max {(x0_ - e0_).max() / 100.0:.2f} us")
        if os.environ.get('STAMP_WGDIST'):
            ra, rb = (x0_ - e0_) / 100.0, (x1_ - e1_) / 100.0
            print("  residence by XCD (block % 8), us, launch A | B: " + "; ".join(f"{k}: {np.median(ra[k::8]):.1f} [{ra[k::8].min():.1f}-{ra[k::8].max():.1f}] | {np.median(rb[k::8]):.1f} [{rb[k::8].min():.1f}-{rb[k::8].max():.1f}]" for k in range(8)))
            print(f"  correlation of a workgroup's residence in A and in B: {np.corrcoef(ra, rb)[0, 1]:.2f}; deciles A: {np.percentile(ra, [0,10,25,50,75,90,100]).round(1).tolist()}; deciles B: {np.percentile(rb, [0,10,25,50,75,90,100]).round(1).tolist()}")
            print(f"  entry by XCD, us after the launch's first entry (A): " + "; ".join(f"{k}: {np.median(e0_[k::8] - e0_.min())/100.0:.2f}" for k in range(8)))
