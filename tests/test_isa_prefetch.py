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


import importlib.util

_spec = importlib.util.spec_from_file_location("check_isa", os.path.join(CSRC, "check_isa.py"))
check_isa = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(check_isa)


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "conv_igemm.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-inline-asm",
           "-S", "--cuda-device-only", os.path.join(CSRC, "conv_igemm.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    return out.read_text()


@pytest.mark.parametrize("em", [1, 2, 3])
@pytest.mark.parametrize("nks", [4, 1])
@pytest.mark.parametrize("dt", [0, 1])
def test_prefetch_registers_untouched_until_wait(asm, dt, nks, em):
    """The same check the Makefile runs as a build gate (csrc/check_isa.py); em: 1 = residual, 2 = mask, 3 = mask sign bits."""
    assert check_isa.check_kernel(asm, dt, nks, em) > 100


def test_gate_rejects_a_touched_register(asm):
    """Plant a read of a prefetch destination between a load and the wait: the gate must fail."""
    name = check_isa.kernel_name(0, 4, 1)
    start = asm.index(name + "iiiijiiiiiii:")
    body = asm[start:asm.index(".Lfunc_end", start)].split("\n")
    k = [i for i, l in enumerate(body) if "buffer_load_dwordx4" in l and " lds" not in l and "ASMSTART" in body[i - 1]][0]
    reg = re.search(r"v\[(\d+):\d+\]", body[k]).group(1)
    body.insert(k + 3, f"\tv_mov_b32_e32 v1, v{reg}")
    bad = asm[:start] + "\n".join(body) + asm[asm.index(".Lfunc_end", start):]
    with pytest.raises(AssertionError):
        check_isa.check_kernel(bad, 0, 4)
