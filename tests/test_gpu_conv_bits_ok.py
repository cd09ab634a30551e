"""srk_conv_bits_ok's refusals (pixel shuffle, post_add)."""


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


def test_conv_bits_ok_refuses_pixel_shuffle_and_post_add(A):
    """the sign-bit store indexes its [N*H*W][2] buffer by the OUTPUT pixel: a pixel-shuffled store (ps_r > 1) or a planar
    epilogue with post_add is not a bits launch -- srk_conv_bits_ok is the gate the C ABI advertises and must say so."""
    L = A._lib
    lib = L.load()
    n, hw = 2, 16
    x = torch.zeros(n, hw, hw, 64, dtype=torch.bfloat16, device="cuda")
    out = torch.empty_like(x)
    w = torch.nn.Parameter(torch.zeros(64, 64, 3, 3, device="cuda"))
    pk = A.ops.pack_conv(w, None, torch.bfloat16)
    bits = torch.empty(n * hw * hw, 2, dtype=torch.int32, device="cuda")

    def args(**kw):
        a = L.ConvArgs(x=x.data_ptr(), x_pitch=64, x_coff=0, N=n, H=hw, W=hw, Cin=64, wpk=pk.wpk.data_ptr(), bias=0, CoutP=64, Cout=64, KH=3, KW=3,
                       relu=1, scale=1.0, res=0, mask=0, out=out.data_ptr(), out_pitch=64, out_coff=0, out_mode=L.OUT_NHWC, ps_r=0, post_add=0,
                       dtype=L.SRK_BF16, relu_bits=bits.data_ptr(), mask_bits=0)
        for k, v in kw.items():
            setattr(a, k, v)
        return a
    import ctypes as C
    assert lib.srk_conv_bits_ok(C.byref(args())) == 1
    assert lib.srk_conv_bits_ok(C.byref(args(ps_r=2, out_mode=L.OUT_NHWC_PS))) == 0
    assert lib.srk_conv_bits_ok(C.byref(args(ps_r=2))) == 0
    assert lib.srk_conv_bits_ok(C.byref(args(post_add=bits.data_ptr()))) == 0


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
