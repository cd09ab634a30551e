#!/bin/bash
# usage: ./resusage.sh file.hip  -> one line per kernel: name vgpr agpr sgpr spill occupancy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith("Function Name:"):
        if cur: print(cur)
        cur={"name":t.split(":",1)[1].strip()[:70]}
    else:
        k,_,v=t.partition(":"); k=k.strip()
        if k in ("VGPRs","AGPRs","TotalSGPRs","VGPRs Spill","SGPRs Spill","Occupancy [waves/SIMD]","ScratchSize [bytes/lane]"): cur[k.split()[0] if k!="VGPRs Spill" and k!="SGPRs Spill" else k]=v.strip()
if cur: print(cur)
'
