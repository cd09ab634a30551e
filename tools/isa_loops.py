#!/usr/bin/env python3
"""Instruction mix of every loop that holds MFMAs in one kernel of an ISA listing (hipcc -S --cuda-device-only).
usage: isa_loops.py file.s kernel-name-substring"""
import re, sys
text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
lines = text[start:end + 1]
lab = {}
for i, l in enumerate(lines):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        lab[m.group(1)] = i
PATS = [("mfma", r'v_mfma'), ("acc_read", r'v_accvgpr_read'), ("acc_write", r'v_accvgpr_write'), ("acc_mov", r'v_accvgpr_mov'),
        ("v_mov", r'v_mov_b'), ("ds_read", r'ds_read'), ("valu", r'^\s+v_(?!mfma|accvgpr|mov)'), ("salu", r'^\s+s_(?!waitcnt|nop|cbranch|branch|barrier)'),
        ("waitcnt", r's_waitcnt'), ("nop", r's_nop'), ("branch", r's_cbranch|s_branch'), ("vmem", r'buffer_|global_'), ("scratch", r'scratch_')]
print(f"{key}: {len(lines)} lines")
for i, l in enumerate(lines):
    m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
    if m and m.group(1) in lab and lab[m.group(1)] < i:
        seg = lines[lab[m.group(1)]:i + 1]
        cnt = {k: sum(1 for x in seg if re.search(p, x)) for k, p in PATS}
        if cnt["mfma"] >= 8:
            print(f"  loop {lab[m.group(1)]}-{i}: " + ", ".join(f"{k} {v}" for k, v in cnt.items()))
