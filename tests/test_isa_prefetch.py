"""Static check of the hand-ordered prefetch in conv_ws_kernel<..., EARLY = true> (no GPU needed: hipcc -S).

The residual / mask pieces are loaded from inline asm, which the compiler treats as an ordinary register definition:
nothing tells it that the registers only become valid at the explicit `s_waitcnt vmcnt(0)` that closes the MFMA
phase.  The kernel is only correct if, in the generated code, (1) no instruction reads or writes a prefetch
destination between that load and the wait, and (2) the path between them is straight-line (no label: a branch
target there could be entered with different registers in flight).  This test compiles the translation unit to
assembly and checks exactly that for both 16-bit types."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "sr-pytorch-lightning_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _regs(line):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", line):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", line):
        out.add(int(m.group(1)))
    return out


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "conv_igemm.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm",
           "-S", "--cuda-device-only", os.path.join(CSRC, "conv_igemm.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


@pytest.mark.parametrize("nks", [4, 1])
@pytest.mark.parametrize("dt", [0, 1])
def test_prefetch_registers_untouched_until_wait(asm, dt, nks):
    name = f"_ZN12_GLOBAL__N_114conv_ws_kernelILi{dt}ELi2ELi{nks}ELb1ELb1EEEv13srk_conv_args"
    start = asm.index(name + "iiiijiiiiiii:")
    body = asm[start:asm.index(".Lfunc_end", start)].split("\n")
    loads = [k for k, l in enumerate(body)
             if "buffer_load_dwordx4" in l and " lds" not in l and "ASMSTART" in body[k - 1]]
    assert len(loads) == 8, "4 pieces x 2 pixel blocks"
    mfma = [k for k, l in enumerate(body) if "v_mfma" in l]
    assert len(mfma) == 36 * nks
    waits = [k for k, l in enumerate(body) if "s_waitcnt vmcnt(0)" in l and "ASMSTART" in body[k - 1] and k > mfma[-1]]
    assert waits, "the closing wait of the MFMA phase"
    wait = waits[0]
    assert mfma[0] < loads[0] and loads[-1] < mfma[-1] < wait, "loads sit inside the MFMA phase, the wait right behind it"
    assert not [k for k in range(loads[0], wait) if body[k].startswith(".LBB")], "straight-line code up to the wait"
    for k in loads:
        dst = _regs(re.search(r"v\[\d+:\d+\]", body[k]).group(0))
        assert len(dst) == 4
        for x in range(k + 1, wait):
            line = body[x].strip()
            if not line or line.startswith((";", ".")):
                continue
            assert not (_regs(line) & dst), f"line {x} touches an in-flight prefetch register: {line}"
    # and the compiler put no wait of its own on the vector-memory counter anywhere in the kernel's loops
    own = [l.strip() for k, l in enumerate(body) if "s_waitcnt vmcnt" in l and "ASMSTART" not in body[k - 1] and k > loads[0]]
    assert not own, own
