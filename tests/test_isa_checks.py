"""Build-time checks of the persistent pointwise-chain kernel's machine code (csrc/pw_chain.hip, pw_fwd2_kernel).

Its MFMAs for the first conv sit in inline asm (the accumulators' register file is chosen by hand) and its `s_waitcnt vmcnt(N)`
are counted by hand over LDS-DMA transfers the compiler does not see.  Two things the compiler can then do silently break it, and
both show in the ISA listing:
  * a register spill: scratch loads / stores are vector-memory instructions the hand counts do not include;
  * a copy, spill or read placed right behind an asm MFMA: the compiler adds the MFMA's wait states only for MFMAs it can see.
hipcc cross-compiles the listing without a GPU (about a minute)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sr-pytorch-lightning_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def listing(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "pw_chain.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", os.path.join(CSRC, "pw_chain.hip"), "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


KERNELS = ["pw_fwd2_kernelILi0ELi8ELi4", "pw_fwd2_kernelILi1ELi8ELi4", "pw_fwd2_kernelILi0ELi4ELi2", "pw_fwd2_kernelILi1ELi4ELi2"]


def test_persistent_forward_has_no_scratch(listing):
    text = open(listing).read()
    for k in KERNELS:
        m = re.search(r"\.name:\s+\S*" + k + r"\S*\n(?:.*\n){0,40}?\s+\.private_segment_fixed_size:\s+(\d+)", text)
        assert m, f"{k}: kernel metadata not found"
        assert int(m.group(1)) == 0, f"{k} uses {m.group(1)} bytes of scratch per lane: spills break its hand-counted vmcnt waits"
        body = text[text.index("\n_ZN12_GLOBAL__N_114" + k):]
        body = body[:body.index("s_endpgm")]
        assert "scratch_" not in body


def test_persistent_forward_asm_mfma_results_are_not_read_early(listing):
    for k in KERNELS:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mfma_hazards.py"), listing, k], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
