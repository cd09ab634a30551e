#!/usr/bin/env python3
"""Where conv_pair_kernel's LDS bank conflicts come from (175,616 SQ_LDS_BANK_CONFLICT per launch at 16 x 48 x 48 =
686 per workgroup, never attributed).  Replays every LDS access site of csrc/conv_pair.hip through the bank model of
tools/lds_model.py, for the ResBlock flavour tools/microbench_pair.py launches (relu_mid, mid, scale_out, res = x).
Usage: python3 tools/lds_conflicts_pair.py [--layout old|new]"""
import sys
from lds_model import extra_cycles, swz

XP = MP = 18
XS, MS = 0, 18 * 18 * 128
WR = MS + 16 * 18 * 128
WG = 3 * 8 * 64 * 16
TO = 14


def site(name, instr, waves, addr_fn, times=1):
    """addr_fn(wave, lane) -> byte address or None; `times` = identical repetitions per workgroup"""
    cyc = ext = n = 0
    for w in waves:
        c, e = extra_cycles(instr, [addr_fn(w, l) for l in range(64)])
        cyc += c * times
        ext += e * times
        n += times
    print(f"{name:58s} {instr:14s} {n:5d} instr  {cyc:6d} cycles  {ext:6d} conflict cycles")
    return ext


def main():
    total = 0
    cw = range(4)
    # 1/2: operand fetches of one conv (36 K-steps x (2 weight + 2 pixel fragments) per compute wave)
    for conv, (base, pitch) in enumerate(((XS, XP), (MS, MP))):
        ext = 0
        for s in range(36):
            kh, kw, ks = s // 12, (s % 12) // 4, s % 4
            for q in range(2):
                ext += site(f"conv{conv+1} weight fragment s={s} q={q}" if s == 0 else "", "ds_read_b128", cw,
                            lambda w, l: WR + ((kh % 3) * WG) + (((l >> 5) * 64 + (l & 31)) << 4) + (((kw * 8 + 2 * ks) * 64 + q * 32) << 4)) if s == 0 else \
                    sum(extra_cycles("ds_read_b128", [WR + ((kh % 3) * WG) + (((l >> 5) * 64 + (l & 31)) << 4) + (((kw * 8 + 2 * ks) * 64 + q * 32) << 4) for l in range(64)])[1] for w in cw)
            for pb in range(2):
                def a(w, l):
                    r, h = l & 31, l >> 5
                    prow, px = 4 * w + 2 * pb + (r >> 4), r & 15
                    return base + ((prow * pitch + px) << 7) + ((kh * pitch + kw) << 7) + (((2 * ks + h) ^ swz(px + kw)) << 4)
                ext += sum(extra_cycles("ds_read_b128", [a(w, l) for l in range(64)])[1] for w in cw)
        print(f"conv{conv+1}: all 36 x 4 operand fetches of the 4 compute waves: {ext} conflict cycles")
        total += ext

    # 3: intermediate epilogue writes (4 per pixel block)
    def mid_w(pb, j):
        def a(w, l):
            r, h = l & 31, l >> 5
            prow, px = 4 * w + 2 * pb + (r >> 4), r & 15
            return MS + ((prow * MP + px) << 7) + (((4 * h + j) ^ swz(px)) << 4)
        return a
    for pb in range(2):
        for j in range(4):
            total += site(f"mid epilogue write pb={pb} j={j}", "ds_write_b128", cw, mid_w(pb, j))

    # 5/6: residual read + result write in the input tile's place
    def out_a(pb, j):
        def a(w, l):
            r, h = l & 31, l >> 5
            prow, px = 4 * w + 2 * pb + (r >> 4), r & 15
            return XS + (((prow + 2) * XP + px + 2) << 7) + (((4 * h + j) ^ swz(px + 2)) << 4)
        return a
    def out_w(pb, j):
        f = out_a(pb, j)
        def a(w, l):
            r = l & 31
            prow, px = 4 * w + 2 * pb + (r >> 4), r & 15
            return f(w, l) if prow < TO and px < TO else None
        return a
    for pb in range(2):
        for j in range(4):
            total += site(f"residual read pb={pb} j={j}", "ds_read_b128", cw, out_a(pb, j))
            total += site(f"output write pb={pb} j={j}", "ds_write_b128", cw, out_w(pb, j))

    # 7: copy-out reads, all 8 waves, 4 per lane
    for k in range(4):
        def a(w, l):
            i = w * 64 + l + 512 * k
            p, c = i >> 3, i & 7
            row, col = divmod(p, TO)
            return XS + (((row + 2) * XP + col + 2) << 7) + ((c ^ swz(col + 2)) << 4)
        total += site(f"copy-out read k={k}", "ds_read_b128", range(8), a)
    # 8: intermediate copy by waves 4..7, 7 per lane
    for k in range(7):
        def a(w, l):
            i = (w - 4) * 64 + l + 256 * k
            p, c = i >> 3, i & 7
            row, col = divmod(p, TO)
            return MS + (((row + 1) * MP + col + 1) << 7) + ((c ^ swz(col + 1)) << 4)
        total += site(f"intermediate copy read k={k}", "ds_read_b128", range(4, 8), a)
    print(f"TOTAL per workgroup: {total} conflict cycles; x 256 workgroups = {total * 256}")


if __name__ == "__main__":
    sys.exit(main())
