#!/usr/bin/env python3
"""pw_wgrad_kernel's LDS access sites through the bank model (tools/lds_model.py): the producers' ds_write_b128 of the h / gh images
and the consumers' transposing reads, with the activation image's swizzle (round 4) and with the images' own swizzle hsw (round 5).
Per 64-pixel tile and workgroup.  Usage: python3 tools/lds_conflicts_pw.py"""
from lds_model import extra_cycles, swz


def hsw(pl):
    return (pl & 1) | (((pl >> 1) & 1) << 2) | (((pl >> 2) & 1) << 1)


def run(name, g):
    wr = rd = 0
    nwr = nrd = 0
    for wave in range(4):                      # producers: pixel block pb = wave & 1, rows rq
        pb = wave & 1
        for u in range(2):
            for m in range(2):
                for img in range(2):           # h, gh
                    rb = 0
                    ad = []
                    for lane in range(64):
                        r, h = lane & 31, lane >> 5
                        pl = 32 * pb + r
                        ad.append((pl << 7) + (((4 * rb + 2 * m + h) ^ g(pl)) << 4))
                    wr += extra_cycles("ds_write_b128", ad)[1]
                    nwr += 1
    for cw in range(4):                        # consumers: per K-step (16 pixels) two reads per operand half
        for rr in range(4):
            for u in range(2):
                for i in range(2):
                    for rdx in range(2):
                        ad = []
                        for lane in range(64):
                            G = lane >> 4
                            hh, rowblk, q, p = G >> 1, G & 1, (lane & 15) >> 2, lane & 3
                            chunk = i * 4 + rowblk * 2 + (p >> 1)
                            col = 8 * hh + 4 * rdx + q
                            ad.append(rr * 2048 + (col << 7) + ((chunk ^ g(col)) << 4) + ((p & 1) << 3))
                        rd += extra_cycles("ds_read_b64_tr_b16", ad)[1]
                        nrd += 1
    print(f"{name:34s}: h/gh stores {nwr} instr, {wr} conflict cycles; transposing reads of h/gh {nrd} instr, {rd} conflict cycles")


run("activation swizzle swz(pl & 15)", lambda pl: swz(pl & 15))
run("own swizzle hsw(pl)", hsw)
print("MFMAs per tile and workgroup: 256")
