#!/usr/bin/env python3
"""Checks one kernel of an ISA listing for instructions that touch the destination registers of a hidden (inline-asm) buffer load
before the kernel's own wait for it: the compiler takes the asm's outputs for defined at once, so any copy or reuse it places
between the load and the hand-written `s_waitcnt vmcnt(N)` (a bare vmcnt wait, which the compiler itself never emits bare in these
kernels) reads or clobbers registers whose data is still in flight.
usage: isa_hidden_loads.py file.s kernel-name-substring     exit status 1 on a finding"""
import re, sys
text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
L = text[start:end + 1]

def regs_in(s):
    out = set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]', s): out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r'\bv(\d+)\b', s): out.add(int(m.group(1)))
    return out

loads = [i for i, l in enumerate(L) if re.search(r'buffer_load_dwordx4 v\[\d+:\d+\], v\d+, s\[\d+:\d+\], 0 offen\s*$', l)]
bad = 0
for i in loads:
    m = re.search(r'buffer_load_dwordx4 v\[(\d+):(\d+)\]', L[i])
    dest = set(range(int(m.group(1)), int(m.group(2)) + 1))
    for k in range(i + 1, len(L)):
        s = L[k].split(';')[0].strip()
        if re.fullmatch(r's_waitcnt vmcnt\(\d+\)', s): break          # the kernel's wait for these loads
        if s.startswith('buffer_load_dwordx4') and k in loads: continue
        if s and not s.endswith(':') and regs_in(s) & dest:
            # the address register of the load itself may be one of its destinations; later loads of the group may share nothing
            bad += 1
            print(f"line {start + k}: `{s[:90]}` touches v{min(dest)}..v{max(dest)} of the hidden load at line {start + i}")
            break
print(f"{key}: {len(loads)} hidden loads, {bad} touched before the wait")
sys.exit(1 if bad else 0)
