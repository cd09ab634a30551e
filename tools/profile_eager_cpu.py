import cProfile, pstats, sys, time, torch
sys.path.insert(0, "/root/repo")
import sr_amd as A

from sr_amd import trainer as T
dev = torch.device("cuda")
m = A.EDSR(scale_factor=4, precision="bf16").to(dev)
batch = T.synthetic_batch(256, 3, 48, 4, 1, dev)
opt = torch.optim.Adam(m.parameters(), fused=True)
def step():
    opt.zero_grad(set_to_none=True)
    loss = m._calculate_losses(img_sr=m(batch["lr"]), img_hr=batch["hr"])["loss"]
    loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("eager ms/step", (time.perf_counter() - t0) / 20 * 1e3)
# CPU-only cost: time to ENQUEUE a step (no sync inside)
t0 = time.perf_counter()
for _ in range(20): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("enqueue ms/step", (t1 - t0) / 20 * 1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
