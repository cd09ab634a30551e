"""The head conv's one-thread-per-pixel unfold and its small 1x1 weight-gradient kernel."""


import os


import sys


import pytest


import torch


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


sys.path.insert(0, ROOT)


pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


def _params(seed, shapes):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.rand(*s, generator=g) - 0.5).cuda()) for s in shapes]


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("shape", [(90, 3, 48, 48), (2, 3, 48, 48), (29, 3, 83, 85), (1, 3, 1, 5)])    # >= 200k pixels: the per-pixel kernel
def test_head_unfold_one_thread_per_pixel_is_exact(A, dtype, shape):
    """The 3-channel 3x3 im2col of the head conv (edsr.py:41-44 with sub_mean, common.py:58-71) has its own kernel (one thread per pixel, 64
    bytes per store group): the values are (x - mean) rounded once to the storage type, zeros outside the image and in channels 27..31, in
    torch's unfold order (channel-major, then kernel row, kernel column)."""
    torch.manual_seed(11)
    x = torch.rand(*shape, device="cuda")
    sub = torch.tensor([0.4488, 0.4371, 0.4040], device="cuda")
    got = A.ops.unfold_raw(x, sub, 3, dtype)
    n, c, h, w = shape
    ref = torch.nn.functional.unfold(x - sub.view(1, 3, 1, 1), 3, padding=1).view(n, 27, h, w).permute(0, 2, 3, 1).to(dtype)
    assert got.shape == (n, h, w, 32)
    assert torch.equal(got[..., :27], ref)
    assert not got[..., 27:].any()
    got0 = A.ops.unfold_raw(x, None, 3, dtype)
    ref0 = torch.nn.functional.unfold(x, 3, padding=1).view(n, 27, h, w).permute(0, 2, 3, 1).to(dtype)
    assert torch.equal(got0[..., :27], ref0)


_WG1X1_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
import sr_amd as A
torch.manual_seed(5)
out = {{}}
for tag, (n, h, w, cin, cout) in {{"head": (7, 48, 48, 32, 64), "ragged": (3, 13, 9, 32, 64), "one": (1, 1, 1, 32, 64), "sq": (2, 24, 24, 64, 64)}}.items():
    for dt in (torch.bfloat16, torch.float16):
        x = (torch.rand(n, h, w, cin, device="cuda") - 0.5).to(dt)
        x[..., 27:] = 0
        dy = (torch.rand(n, h, w, cout, device="cuda") - 0.5).to(dt)
        creal = 27 if cin == 32 else cin
        gw, gb = A.ops.wgrad_raw(x, dy, N=n, H=h, W=w, Cin=cin, Cout=cout, k=1, w_shape=(cout, creal, 1, 1), want_bias=True)
        ref = torch.einsum("nhwi,nhwo->oi", x.double(), dy.double())[:, :creal]
        out[f"{{tag}}_{{dt}}"] = (gw.cpu(), gb.cpu(), ref.cpu(), dy.double().sum((0, 1, 2)).cpu())
torch.save(out, sys.argv[1])
"""


def test_small_1x1_weight_gradient_kernel_matches_the_general_one_bit_for_bit(A, tmp_path):
    """The head conv's weight gradient (K = 27 (32) x 64) runs on a 64 x 64 kernel with a ring of 8 tile buffers instead of the 128 x 256
    one with two (conv_wgrad.hip): same tiles, same MFMA order -- the gradients must be the SAME BITS as the general kernel's (knob
    SRK_NO_WGRAD1X1_SMALL, read once per process: two processes), and right against float64."""
    import subprocess
    res = {}
    for knob in ("0", "1"):
        f = tmp_path / f"wg{knob}.pt"
        env = dict(os.environ, SRK_DEBUG="1")
        if knob == "1":
            env["SRK_NO_WGRAD1X1_SMALL"] = "1"
        else:
            env.pop("SRK_NO_WGRAD1X1_SMALL", None)
        out = subprocess.run([sys.executable, "-c", _WG1X1_SCRIPT.format(root=ROOT), str(f)], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout + out.stderr
        res[knob] = torch.load(f)
    assert res["0"].keys() == res["1"].keys() and len(res["0"]) == 8
    for k in res["0"]:
        gw0, gb0, ref, refb = res["0"][k]
        gw1, gb1, _, _ = res["1"][k]
        assert torch.equal(gw0, gw1) and torch.equal(gb0, gb1), k
        scale = ref.abs().max().item() + 1e-9
        assert (gw0.double().flatten(1) - ref).abs().max().item() <= 2e-5 * scale + 1e-6, k
        assert (gb0.double() - refb).abs().max().item() <= 2e-5 * (refb.abs().max().item() + 1.0), k
