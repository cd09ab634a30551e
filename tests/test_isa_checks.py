"""Build-time checks of the machine code of the two persistent kernels with hand-counted vector-memory waits: the pointwise chain's
forward (csrc/pw_chain.hip, pw_fwd2_kernel) and the K-streaming 3x3 conv (csrc/conv_ks.hip, conv_ks_kernel).

Its MFMAs for the first conv sit in inline asm (the accumulators' register file is chosen by hand) and its `s_waitcnt vmcnt(N)`
are counted by hand over LDS-DMA transfers the compiler does not see.  Two things the compiler can then do silently break it, and
both show in the ISA listing:
  * a register spill: scratch loads / stores are vector-memory instructions the hand counts do not include;
  * a copy, spill or read placed right behind an asm MFMA: the compiler adds the MFMA's wait states only for MFMAs it can see.
  * (conv_ks) anything touching the destination registers of a hidden buffer load before the kernel's own wait for it.
hipcc cross-compiles the listings without a GPU (about a minute)."""
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
        body = body[:body.index(".Lfunc_end")]
        assert "scratch_" not in body


def test_persistent_forward_asm_mfma_results_are_not_read_early(listing):
    for k in KERNELS:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_mfma_hazards.py"), listing, k], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]


@pytest.fixture(scope="module")
def listing_ks(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "conv_ks.s")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", os.path.join(CSRC, "conv_ks.hip"), "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return out


KS_KERNELS = ["conv_ks_kernelILi0ELb0", "conv_ks_kernelILi0ELb1", "conv_ks_kernelILi1ELb0", "conv_ks_kernelILi1ELb1"]


def test_conv_ks_has_no_scratch_and_its_stream_is_straight_line(listing_ks):
    text = open(listing_ks).read()
    for k in KS_KERNELS:
        m = re.search(r"\.name:\s+\S*" + k + r"\S*\n(?:.*\n){0,40}?\s+\.private_segment_fixed_size:\s+(\d+)", text)
        assert m, f"{k}: kernel metadata not found"
        assert int(m.group(1)) == 0, f"{k} uses {m.group(1)} bytes of scratch per lane: spills break its hand-counted vmcnt waits"
        body = text[text.index("\n_ZN12_GLOBAL__N_114" + k):]
        body = body[:body.index(".Lfunc_end")]
        assert "scratch_" not in body
        # the K-block's 36 steps x 4 MFMAs exist once, as straight-line code (a rolled loop would index the fragment buffers dynamically)
        assert body.count("v_mfma_f32_32x32x16") == 144, f"{k}: {body.count('v_mfma_f32_32x32x16')} MFMAs"


def test_conv_ks_hidden_loads_are_not_touched_before_their_wait(listing_ks):
    for k in KS_KERNELS:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_hidden_loads.py"), listing_ks, k], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-2000:]


# ---- the round-3 kernels that keep a whole weight tensor or K accumulator tiles in registers (csrc/proj.hip, csrc/conv_lk.hip): a register
# ---- spill there is not a correctness problem (their waits are full `vmcnt(0)` waits or the compiler's own) but a silent 2x slowdown
@pytest.mark.parametrize("src,limits", [
    ("proj.hip", {"proj_down_kernelILi0E": 16, "proj_down_kernelILi1E": 16, "proj_up_kernelILi0E": 32, "proj_up_kernelILi1E": 96,       # (a few spilled pointers; the fp16 conversions cost the fused-PReLU epilogue more)
                  "proj_wgrad_kernelILi0E": 0, "proj_wgrad_kernelILi1E": 0}),
    ("conv_lk.hip", {"lk_wgrad_allrows_kernelILi0ELi9E": 0, "lk_wgrad_packed_kernelILi0ELi9E": 0, "lk_conv_rows_kernelILi0ELi9E": 0,
                     "lk_wgrad_allrows_kernelILi1ELi9E": 0, "lk_conv_rows_kernelILi1ELi9E": 0,
                     # the collapsed HR stage's 5x5 kernels: weights stationary in 160 / 200 registers AND hand-counted vmcnt waits behind hidden
                     # LDS-DMA -- a spill there is a scratch access the counts do not include (a correctness problem, not only a slow-down)
                     "lk5_rows_fwd_kernelILi0E": 0, "lk5_rows_fwd_kernelILi1E": 0, "lk5_dgrad_kernelILi0E": 0, "lk5_dgrad_kernelILi1E": 0,
                     "lk5_fwd_kernelILi0E": 0, "lk5_fwd_kernelILi1E": 0, "lk5_wgrad_kernelILi0E": 0, "lk5_wgrad_kernelILi1E": 0}),
])
def test_register_resident_kernels_do_not_spill(tmp_path, src, limits):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    out = str(tmp_path / (src + ".s"))
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                        "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    text = open(out).read()
    for k, lim in limits.items():
        m = re.search(r"\.name:\s+\S*" + k + r"\S*\n(?:.*\n){0,40}?\s+\.private_segment_fixed_size:\s+(\d+)", text)
        assert m, f"{k}: kernel metadata not found"
        assert int(m.group(1)) <= lim, f"{k} uses {m.group(1)} bytes of scratch per lane (allowed: {lim})"
