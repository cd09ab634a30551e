#!/usr/bin/env python3
"""Build gate: static check of the hand-ordered prefetch in conv_ws_kernel<..., EARLY = true> on the generated ISA.

The residual / mask pieces of the prefetch variant are loaded from inline asm, which the compiler treats as an ordinary
register definition: nothing tells it that the registers only become valid at the explicit `s_waitcnt vmcnt(0)` that
closes the MFMA phase.  The kernel is correct only if, in the code hipcc actually generated, (1) no instruction reads
or writes a prefetch destination between that load and the wait, (2) the path between them is straight-line (no
label: a branch target there could be entered with other registers in flight) and (3) the compiler placed no vmcnt
wait of its own behind the first prefetch.  A toolchain upgrade that breaks any of these fails the BUILD (the Makefile
runs this on every change of conv_igemm.hip); `SRK_NO_EARLY=1` at run time routes those launches to the plain variant.

usage: check_isa.py conv_igemm.s   (exit code 0 = all instantiations pass)"""
import re
import sys


def _regs(line):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


def kernel_name(dt, nks, em=1):
    """Mangled name of conv_ws_kernel<dt, 2, nks, FAST = true, EARLY = true, EM = em> (em: 1 = residual, 2 = mask prefetch)."""
    return f"_ZN12_GLOBAL__N_114conv_ws_kernelILi{dt}ELi2ELi{nks}ELb1ELb1ELi{em}EEEv13srk_conv_args"


def check_kernel(asm, dt, nks, em=1):
    """Raises AssertionError with the offending line; returns the number of instructions inspected."""
    name = kernel_name(dt, nks, em)
    start = asm.index(name + "iiiijiiiiiii:")
    body = asm[start:asm.index(".Lfunc_end", start)].split("\n")
    if em == 3:       # sign bits: one 4-byte word per pixel block
        loads = [k for k, l in enumerate(body) if re.search(r"buffer_load_dword\s", l) and " lds" not in l and "ASMSTART" in body[k - 1]]
        assert len(loads) == 2, f"expected one word x 2 pixel blocks, found {len(loads)}"
    else:
        loads = [k for k, l in enumerate(body)
                 if "buffer_load_dwordx4" in l and " lds" not in l and "ASMSTART" in body[k - 1]]
        assert len(loads) == 8, f"expected 4 pieces x 2 pixel blocks, found {len(loads)}"
    mfma = [k for k, l in enumerate(body) if "v_mfma" in l]
    assert len(mfma) == 36 * nks, f"{len(mfma)} MFMAs"
    waits = [k for k, l in enumerate(body) if "s_waitcnt vmcnt(0)" in l and "ASMSTART" in body[k - 1] and k > mfma[-1]]
    assert waits, "the closing wait of the MFMA phase is missing"
    wait = waits[0]
    assert mfma[0] < loads[0] and loads[-1] < mfma[-1] < wait, "loads must sit inside the MFMA phase, the wait right behind it"
    assert not [k for k in range(loads[0], wait) if body[k].startswith(".LBB")], "branch target between a prefetch and its wait"
    seen = 0
    for k in loads:
        if em == 3:
            dst = _regs(re.search(r"buffer_load_dword\s+(v\d+)", body[k]).group(1))
            assert len(dst) == 1
        else:
            dst = _regs(re.search(r"v\[\d+:\d+\]", body[k]).group(0))
            assert len(dst) == 4
        for x in range(k + 1, wait):
            line = body[x].strip()
            if not line or line.startswith((";", ".")):
                continue
            seen += 1
            assert not (_regs(line) & dst), f"line {x} touches an in-flight prefetch register: {line}"
    own = [l.strip() for k, l in enumerate(body) if "s_waitcnt vmcnt" in l and "ASMSTART" not in body[k - 1] and k > loads[0]]
    assert not own, f"compiler-placed vmcnt wait behind the prefetch: {own}"
    return seen


def isa_fingerprint(asm, name):
    """sha256 (16 hex digits) of the generated ISA of ONE kernel, comments / blank lines / block-label numbering removed: what a PMC
    measurement of that kernel belongs to (bench.py quotes `roofline.traffic` only for the fingerprint it was collected on).  A change
    anywhere else in the translation unit, or in a header the kernel does not use, leaves it alone."""
    import hashlib
    start = asm.index(name)
    body = asm[start:asm.index(".Lfunc_end", start)].split("\n")[1:]
    keep = []
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") and not l.startswith(".LBB"):
            continue
        keep.append(re.sub(r"\.LBB\d+_", ".LBB_", l))
    return hashlib.sha256("\n".join(keep).encode()).hexdigest()[:16]


# the kernels whose PMC figures bench.py quotes: key -> mangled-name prefix
FINGERPRINTED = {
    "conv_ws_plain_bf16": "_ZN12_GLOBAL__N_114conv_ws_kernelILi0ELi2ELi4ELb1ELb0ELi0EEEv13srk_conv_argsiiiijiiiiiii:",
    "conv_ws_residual_bf16": kernel_name(0, 4, 1) + "iiiijiiiiiii:",
    "conv_ws_mask_bf16": kernel_name(0, 4, 2) + "iiiijiiiiiii:",
    "conv_ws_maskbits_bf16": kernel_name(0, 4, 3) + "iiiijiiiiiii:",
    "conv_trunk_bf16": "_ZN12_GLOBAL__N_117conv_trunk_kernelILi0EEEvPK13srk_conv_argsiiiij:",
}


# kernels of the OTHER translation units whose PMC figures bench.py quotes for configs 3-5 (tools/pmc_traffic.sh): file stem -> {key: mangled-name prefix}
FINGERPRINTED_MORE = {
    "conv_pair": {"conv_pair_bf16": "_ZN12_GLOBAL__N_116conv_pair_kernelILi0EEEv18srk_conv_pair_argsiij:"},
    "conv_ks": {"conv_ks_bf16": "_ZN12_GLOBAL__N_114conv_ks_kernelILi0ELb0EEEv13srk_conv_argsiiijji:",
                "conv_ks_psdgrad_bf16": "_ZN12_GLOBAL__N_114conv_ks_kernelILi0ELb1EEEv13srk_conv_argsiiijji:"},
    "pw_chain": {"pw_fwd_bf16": "_ZN12_GLOBAL__N_113pw_fwd_kernelILi0ELi8ELi4EEEv11srk_pw_argsjji:",
                 "pw_bwd_bf16": "_ZN12_GLOBAL__N_113pw_bwd_kernelILi0ELi8ELi4ELb0EEEv15srk_pw_bwd_argsjjji:",
                 "pw_wgrad_bf16": "_ZN12_GLOBAL__N_115pw_wgrad_kernelILi0ELi8ELi4EEEv17srk_pw_wgrad_argsjjiii:"},
    "proj": {"proj_up_bf16": "_ZN12_GLOBAL__N_114proj_up_kernelILi0EEEv13srk_proj_argsiii:"},
}


def main(path, fp_out=None, more=()):
    """path: conv_igemm's listing (prefetch gate + fingerprints); more: listings of other translation units (fingerprints only)."""
    asm = open(path).read()
    ok = True
    for dt in (0, 1):
        for nks in (4, 1):
            for em in (1, 2, 3):
                try:
                    n = check_kernel(asm, dt, nks, em)
                    print(f"check_isa: conv_ws_kernel<dtype {dt}, NKS {nks}, EARLY, EM {em}>: ok ({n} instructions between prefetch and wait)")
                except (AssertionError, ValueError) as e:
                    ok = False
                    print(f"check_isa: conv_ws_kernel<dtype {dt}, NKS {nks}, EARLY, EM {em}>: FAILED: {e}", file=sys.stderr)
    if fp_out:
        import json
        import os
        tab = {k: isa_fingerprint(asm, v) for k, v in FINGERPRINTED.items()}
        for m in more:
            stem = os.path.basename(m).split(".")[0]
            other = open(m).read()
            for k, v in FINGERPRINTED_MORE.get(stem, {}).items():
                tab[k] = isa_fingerprint(other, v)
        with open(fp_out, "w") as fh:
            json.dump(tab, fh, indent=1)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3:]))
