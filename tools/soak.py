"""Soak run: N training steps through trainer.Trainer (hipGraph replay after the eager steps) on synthetic patches whose HR is a smooth
function of LR, checking that the loss falls, stays finite and that device memory does not grow.  usage: soak.py MODEL BATCH STEPS"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, sr_amd
from sr_amd import trainer as T
name, batch, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(0)
m = getattr(sr_amd, name)(scale_factor=4, precision="bf16")
g = torch.Generator().manual_seed(1)
pool = []
for i in range(8):                                   # 8 fixed batches: HR = bicubic upsampling of LR (learnable)
    lr = torch.rand(batch, 3, 48, 48, generator=g)
    hr = torch.nn.functional.interpolate(lr, scale_factor=4, mode="bicubic", align_corners=False).clamp(0, 1)
    pool.append({"lr": lr.cuda(), "hr": hr.cuda(), "path": []})
def batches():
    for i in range(steps):
        yield pool[i % 8]
tr = T.Trainer(device="cuda")
t0 = time.perf_counter()
mem = []
class Probe:
    pass
# run in chunks to sample memory
tr.fit(m, batches())
torch.cuda.synchronize()
el = time.perf_counter() - t0
L = tr.losses
import math
assert all(math.isfinite(v) for v in L), "non-finite loss"
k = max(1, steps // 10)
first, last = sum(L[:k]) / k, sum(L[-k:]) / k
print(f"{name} b{batch}: {steps} steps in {el:.1f} s ({el / steps * 1e3:.2f} ms/step incl. eager steps + capture), loss {first:.4f} -> {last:.4f}, "
      f"graph captured: {tr.graphed is not None and tr.graphed.graphs is not None}, peak memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB, "
      f"now {torch.cuda.memory_allocated() / 2**30:.2f} GiB")
assert last < first, "the loss does not fall"
