import os, sys, torch
sys.path.insert(0, "/root/repo")
import sr_amd as A
dev=torch.device("cuda"); dt=torch.bfloat16
n=64
x=(torch.rand(n,192,192,16,device=dev)-0.5).to(dt)
w=torch.nn.Parameter((torch.rand(64,16,3,3,device=dev)-0.5)*0.05); b=torch.nn.Parameter(torch.zeros(64,device=dev))
pk=A.ops.pack_conv(w,b,dt)
out=torch.empty(n,192,192,64,device=dev,dtype=dt)
f=lambda: A.ops.conv_raw(x,pk,N=n,H=192,W=192,Cin=16,Cout=64,out=out)
for _ in range(3): f()
torch.cuda.synchronize()
g=torch.cuda.CUDAGraph()
st=torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(st): f()
torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
with torch.cuda.graph(g):
    for _ in range(10): f()
g.replay(); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
print("tail dgrad 16->64 @192x192 n=%d: %.1f us  (lib %s)"%(n, e0.elapsed_time(e1)*100, os.environ.get("SRK_LIB_PATH","current")))
