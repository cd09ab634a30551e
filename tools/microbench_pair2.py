"""Two half-batch conv_pair chains on two streams of one hipGraph against one whole-batch chain (round 5 experiment: does a second
stream hide the launch boundary of a chain of dependent launches?).  Usage: python3 tools/microbench_pair2.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sr_amd as A

dev = torch.device("cuda"); dt = torch.bfloat16
A._lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = 32
x = (torch.rand(n, 48, 48, 64, device=dev) - 0.5).to(dt)
ws = [torch.nn.Parameter((torch.rand(64, 64, 3, 3, device=dev) - 0.5) * 0.05) for _ in range(2)]
pk = [A.ops.pack_conv(w, torch.nn.Parameter(torch.zeros(64, device=dev)), dt) for w in ws]
mid, o1, o2 = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)

def chain(lo, hi):
    a, o = x[lo:hi], o1[lo:hi]
    for _ in range(L):
        A.ops.conv_pair_raw(a, pk[0], pk[1], out=o, relu_mid=True, mid=mid[lo:hi], scale_out=0.1, res=a)
        a, o = o, (o2[lo:hi] if o.data_ptr() == o1[lo:hi].data_ptr() else o1[lo:hi])

def timed(capture, label):
    s1 = torch.cuda.Stream(); s1.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s1):
        capture(s1, warm=True)
    torch.cuda.current_stream().wait_stream(s1); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s1):
        capture(s1, warm=False)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"N={n} {label}: {e0.elapsed_time(e1) * 1000 / 20 / L:.2f} us per block of the whole batch")

def one(s1, warm):
    chain(0, n)

s2 = torch.cuda.Stream()
def two(s1, warm):
    ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
    with torch.cuda.stream(s2):
        chain(n // 2, n)
    chain(0, n // 2)
    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)

def seq(s1, warm):
    chain(0, n // 2); chain(n // 2, n)

timed(one, "one chain, whole batch            ")
timed(two, "two half-batch chains, two streams")
timed(seq, "two half-batch chains, one stream ")
