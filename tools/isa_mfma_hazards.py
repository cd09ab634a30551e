#!/usr/bin/env python3
"""Checks one kernel of an ISA listing (hipcc -S --cuda-device-only) for reads of an inline-asm MFMA's result that come too early.
The compiler inserts the wait states an MFMA needs only for MFMAs it can see; one inside `asm` is opaque, so every copy, spill or
VALU / LDS / memory read the register allocator or the scheduler places behind it is unprotected.  An 8-pass MFMA's result may be
read by a non-MFMA instruction 11 wait states (instruction issues / s_nop counts) after the MFMA at the earliest; two further
8-pass MFMAs issued in between also cover it (each holds issue for its passes).
usage: isa_mfma_hazards.py file.s kernel-name-substring     exit status 1 if a hazard is found"""
import re, sys

def regs(tok):
    m = re.match(r'^([va])\[(\d+):(\d+)\]$', tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'^([va])(\d+)$', tok)
    return {(m.group(1), int(m.group(2)))} if m else set()

text = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(text) if l.startswith('_Z') and key in l.split(':')[0] and ':' in l)
end = next(i for i in range(start, len(text)) if text[i].startswith('.Lfunc_end'))
lines = text[start:end + 1]
in_asm, bad, pending = False, 0, []          # pending: [dst regs, wait states seen, mfmas seen, line]
for n, l in enumerate(lines):
    if '#ASMSTART' in l: in_asm = True; continue
    if '#ASMEND' in l: in_asm = False; continue
    s = l.split(';')[0].strip()
    if not s or s.endswith(':') or s.startswith('.'):
        if s.endswith(':'): pending = []       # a label: other paths join here; the checker stays inside straight-line code
        continue
    op, _, rest = s.partition(' ')
    toks = [t.strip() for t in rest.split(',')]
    if op.startswith('v_mfma'):
        for p in pending: p[2] += 1
        pending = [p for p in pending if p[2] < 2 and p[1] < 11]
        if in_asm: pending.append([regs(toks[0]), 0, 0, n])
        continue
    used = set()
    for t in toks:
        for w in re.findall(r'[va]\[\d+:\d+\]|\b[va]\d+\b', t): used |= regs(w)
    for p in pending:
        if used & p[0]:
            bad += 1
            print(f"line {start + n}: `{s}` touches the result of the asm MFMA at line {start + p[3]} after {p[1]} wait states, {p[2]} MFMAs")
    ws = 1 + (int(toks[0]) if op == 's_nop' and toks[0].isdigit() else 0)
    for p in pending: p[1] += ws
    pending = [p for p in pending if p[1] < 11]
    if op.startswith('s_cbranch') or op in ('s_branch', 's_barrier'): pending = [] if op != 's_barrier' else pending
print(f"{key}: {bad} early reads of asm MFMA results")
sys.exit(1 if bad else 0)
