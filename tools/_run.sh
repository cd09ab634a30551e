tools/profile_step.sh wdsr_b 16 r5 > /dev/null 2>&1
python3 tools/step_list.py gpurun_out/r5_step_wdsr_b_b16.csv.gz 25 > gpurun_out/r5_step_wdsr_b_b16.txt
python3 - <<'PY'
import collections
agg=collections.OrderedDict()
for l in open('gpurun_out/r5_step_wdsr_b_b16.txt'):
    l=l.rstrip('\n')
    if l.startswith('sum of'): print(l); continue
    name=l[:64].strip(); rest=l[64:].split()
    try: d=float(rest[1])
    except: continue
    a=agg.setdefault(name[:60],[0,0.0]); a[0]+=1; a[1]+=d
for k,(c,d) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:20]:
    print(f"{k:62s} x{c:3d} {d:8.1f}  {d/c:7.1f}")
PY
