"""The image-stationary trunk launch (srk_conv_trunk, ops.ResTrunkFn: EDSR's body as ONE launch per direction, models/edsr.py:24-31,44-47)
against the per-layer launches it replaces -- bit for bit, forward, input gradient and every parameter gradient -- and against the CPU
oracle.  The trunk path needs a batch that is a whole number of rounds over the CUs, so the images are small."""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def A():
    import sr_amd
    return sr_amd


def _cus(A):
    return A._lib.load().srk_device_cus()


def _body(A, nb, dt, seed=0, scale=0.1):
    torch.manual_seed(seed)
    m = A.EDSR(n_feats=64, n_resblocks=nb, res_scale=scale, scale_factor=2, precision={torch.bfloat16: "bf16", torch.float16: 16}[dt]).cuda()
    with torch.no_grad():
        for p in m.body.parameters():
            p.mul_(3.0)                    # default init keeps the residual branch tiny; make it matter
    return m


def _run_body(A, m, f, g, trunk):
    ops = A.ops
    prev = ops._TRUNK_OFF
    ops._TRUNK_OFF = not trunk
    launched = []
    real = ops._trunk_launch

    def spy(layers, dev):
        ok = real(layers, dev)
        launched.append((len(layers), ok))
        return ok
    ops._trunk_launch = spy
    try:
        for p in m.parameters():
            p.grad = None
        x = f.clone().requires_grad_(True)
        with ops.forward_scope(m._pack_group()):
            r = m.body_nhwc(x)
        (r.float() * g.float()).sum().backward()
        torch.cuda.synchronize()
        grads = {k: p.grad.detach().clone() for k, p in m.body.named_parameters()}
        return r.detach().clone(), x.grad.detach().clone(), grads, launched
    finally:
        ops._TRUNK_OFF = prev
        ops._trunk_launch = real


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rounds,h,w,nb", [(1, 16, 16, 2), (1, 20, 28, 3), (2, 9, 33, 1)])
def test_trunk_equals_the_per_layer_launches_bit_for_bit(A, dt, rounds, h, w, nb):
    n = rounds * _cus(A)
    m = _body(A, nb, dt)
    gen = torch.Generator().manual_seed(5)
    f = (torch.rand(n, h, w, 64, generator=gen) - 0.5).to(dt).cuda()
    g = (torch.rand(n, h, w, 64, generator=gen) - 0.5).to(dt).cuda()
    r1, gx1, gr1, l1 = _run_body(A, m, f, g, trunk=True)
    r0, gx0, gr0, l0 = _run_body(A, m, f, g, trunk=False)
    assert l1 == [(2 * nb + 1, True), (2 * nb + 2, True)], l1          # one launch forward, one backward (2 nb + 1 data gradients + the skip's add)
    assert l0 == []
    assert torch.equal(r1, r0)
    assert torch.equal(gx1, gx0)
    for k in gr0:
        assert torch.equal(gr1[k], gr0[k]), k
    assert float(r1.float().abs().mean()) > 0.05 and float(gx1.float().abs().mean()) > 0.05


def test_trunk_without_gradients_rotates_three_buffers_and_matches(A):
    n, dt = _cus(A), torch.bfloat16
    m = _body(A, 4, dt)
    f = (torch.rand(n, 12, 20, 64) - 0.5).to(dt).cuda()
    g = torch.ones_like(f)
    r1, *_ = _run_body(A, m, f, g, trunk=True)
    with torch.no_grad(), A.ops.forward_scope(m._pack_group()):
        before = torch.cuda.memory_allocated()
        r2 = m.body_nhwc(f)
        torch.cuda.synchronize()
        grown = torch.cuda.memory_allocated() - before
    assert torch.equal(r1, r2)
    assert grown <= 1.5 * f.numel() * f.element_size(), grown          # only the result stays allocated


def test_batches_that_do_not_fill_the_chip_keep_the_per_layer_path(A):
    dt = torch.bfloat16
    m = _body(A, 2, dt)
    for n in (1, 16, _cus(A) - 1, _cus(A) + 3):
        f = (torch.rand(n, 8, 8, 64) - 0.5).to(dt).cuda()
        *_, launched = _run_body(A, m, f, torch.ones_like(f), trunk=True)
        assert launched == [], (n, launched)


def test_trunk_table_validation(A):
    """srk_conv_trunk_ok / srk_conv_trunk refuse what the kernel cannot serve (error behaviour of the C-ABI)."""
    L = A._lib
    lib = L.load()
    n, dt = _cus(A), torch.bfloat16
    x = torch.zeros(n, 8, 8, 64, dtype=dt, device="cuda")
    o = torch.empty_like(x)
    w = torch.nn.Parameter(torch.zeros(64, 64, 3, 3, device="cuda"))
    pk = A.ops.pack_conv(w, None, dt)
    good = A.ops._trunk_layer(x, pk, o)
    arr = (L.ConvArgs * 1)(good)
    assert lib.srk_conv_trunk_ok(arr, 1) == 1
    for field, val in (("KH", 5), ("Cin", 128), ("dtype", 2), ("ps_r", 2), ("N", n - 1)):
        bad = A.ops._trunk_layer(x, pk, o)
        setattr(bad, field, val)
        arr = (L.ConvArgs * 1)(bad)
        assert lib.srk_conv_trunk_ok(arr, 1) == 0, field
        tab = torch.zeros(C.sizeof(L.ConvArgs), dtype=torch.uint8, device="cuda")
        assert lib.srk_conv_trunk(arr, tab.data_ptr(), 1, torch.cuda.current_stream().cuda_stream) != 0
        assert b"srk_conv_trunk" in lib.srk_last_error()


def test_trunk_vs_oracle_fp32_reference(A):
    """The trunk path's forward and gradients against the CPU oracle (fp32) on a small body: 16-bit storage tolerance."""
    n, dt, nb, h, w = _cus(A), torch.bfloat16, 2, 8, 8
    m = _body(A, nb, dt, seed=3)
    gen = torch.Generator().manual_seed(9)
    f = (torch.rand(n, h, w, 64, generator=gen) - 0.5).to(dt)
    g = (torch.rand(n, h, w, 64, generator=gen) - 0.5).to(dt)
    r, gx, grads, launched = _run_body(A, m, f.cuda(), g.cuda(), trunk=True)
    assert launched and all(ok for _, ok in launched)
    # oracle: the same arithmetic in fp32 on NCHW (models/edsr.py:44-47, models/common.py:74-109)
    sd = {k: v.detach().float().cpu().clone().requires_grad_(True) for k, v in m.body.state_dict().items()}
    x = f.float().permute(0, 3, 1, 2).clone().requires_grad_(True)
    t = x
    for b in range(nb):
        hmid = torch.relu(torch.nn.functional.conv2d(t, sd[f"{b}.body.0.weight"], sd[f"{b}.body.0.bias"], padding=1))
        t = t + m.body[b].res_scale * torch.nn.functional.conv2d(hmid, sd[f"{b}.body.2.weight"], sd[f"{b}.body.2.bias"], padding=1)
    ref = torch.nn.functional.conv2d(t, sd[f"{nb}.weight"], sd[f"{nb}.bias"], padding=1) + x
    (ref * g.float().permute(0, 3, 1, 2)).sum().backward()
    got = r.float().cpu().permute(0, 3, 1, 2)
    assert float((got - ref.detach()).abs().max()) <= 2e-2 * float(ref.detach().abs().max())
    gxr = x.grad
    assert float((gx.float().cpu().permute(0, 3, 1, 2) - gxr).abs().max()) <= 3e-2 * float(gxr.abs().max())
    for k, gr in grads.items():
        refg = sd[k].grad
        num = float((gr.float().cpu() - refg).norm())
        assert num <= 6e-2 * float(refg.norm()) + 1e-6, (k, num, float(refg.norm()))      # bf16 activations and gradients through 2 blocks (the per-layer path: the same bits)
