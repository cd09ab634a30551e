#!/usr/bin/env python3
"""Diagnostic: s_memtime stamps of workgroup 0 (waves 0 and 4) of pw_fwd_kernel (csrc/pw_chain.hip).  Needs the stamp build:
`make -C sr-pytorch-lightning_amd/csrc stamp`.  usage: stamp_pw.py [batch]"""
import ctypes as C, os, sys
os.environ.setdefault("SRK_LIB_PATH", os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libsrk_stamp.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ops, dt, dev = A.ops, torch.bfloat16, torch.device("cuda")
f, chid, cmid = 128, 768, 102
x = (torch.rand(n, 48, 48, f, device=dev) - 0.5).to(dt)
w1 = torch.nn.Parameter((torch.rand(chid, f, 1, 1, device=dev) - 0.5) * 0.2); b1 = torch.zeros(chid, device=dev)
w2 = torch.nn.Parameter((torch.rand(cmid, chid, 1, 1, device=dev) - 0.5) * 0.1); b2 = torch.zeros(cmid, device=dev)
pk = ops.pw_pack(w1, b1, w2, b2, dt)
z = torch.empty(n, 48, 48, 112, device=dev, dtype=dt)
for _ in range(3): ops.pw_forward_raw(x, pk, z)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): ops.pw_forward_raw(x, pk, z)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print(f"pw_fwd n={n}: {e0.elapsed_time(e1) * 1e3 / 20:.2f} us per launch")
buf = (C.c_ulonglong * 128)()
lib = A._lib.load()
assert lib.srk_pw_read_stamps(buf) == 0
old = os.environ.get("SRK_PW_FWD_OLD")
for w in (0, 1):
    t = list(buf[w * 64:(w + 1) * 64])
    t0 = t[0]
    if old:
        names = {1: "DMA issued", 2: "own pieces landed", 3: "barrier", 4: "slice-0 conv 1 done (loop starts)", 5: "loop done", 6: "stores issued"}
        print(("wave 0: " if w == 0 else "wave 4: ") + "; ".join(f"{names[i]} {t[i] - t0}" for i in (1, 2, 3, 4, 5, 6)))
        print("   per slice [barrier wait, DMA issue, MFMA stream]: " + " | ".join(
            f"{s}: {t[9 + 4 * s] - t[8 + 4 * s]}" f",{t[10 + 4 * s] - t[9 + 4 * s]},{t[11 + 4 * s] - t[10 + 4 * s]}" f" (@{t[8 + 4 * s] - t0})" for s in range(12) if t[11 + 4 * s]))
    else:
        print(("wave 0: " if w == 0 else "wave 2: ") + f"start -> first conv 1 begins {t[1] - t0}, ends {t[2] - t0}; kernel end {t[3] - t0}")
        print("   second tile, per iteration [barrier wait, burst + descriptor, stream] (@ since kernel start): " + " | ".join(
            f"{s}: {t[5 + 4 * s] - t[4 + 4 * s]},{t[6 + 4 * s] - t[5 + 4 * s]},{t[7 + 4 * s] - t[6 + 4 * s]} (@{t[4 + 4 * s] - t0})" for s in range(12) if t[7 + 4 * s]))
        print(f"   third tile starts @{t[52] - t0}")
