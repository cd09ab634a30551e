"""The tiled grouped weight pack is bit-identical to the stand-alone pack."""


import numpy as np


import pytest


import torch


from oracle import functional as OF, train as OT


pytestmark = pytest.mark.gpu


PREC = {torch.float16: 16, torch.bfloat16: "bf16"}


@pytest.fixture(scope="module")
def A():
    import sr_amd
    assert torch.cuda.is_available()
    sr_amd._lib.load()
    return sr_amd


RCAN_KW = dict(n_feats=64, n_resgroups=2, n_resblocks=3, reduction=16, scale_factor=2)


def test_tiled_group_pack_is_bit_identical_to_the_standalone_pack(A):
    """ops.PackGroup.refresh (srk_pack_conv_weights_group_tiled: one 16 x 64-channel tile of one parameter per block, read through
    LDS) against srk_pack_conv_weights on the same parameters: every layout a model uses -- forward / dgrad, 3x3 / 1x1 / 5x5 (the
    latter through the 16-block fallback), channel counts that are not multiples of 8, 16 or 64, the PixelShuffle permutation on a
    forward layout (models/common.py Upsampler) and on a dgrad layout (fallback), bf16 and fp16 -- bit for bit, biases included,
    after the parameters changed in place."""
    from sr_amd import ops
    torch.manual_seed(11)
    shapes = [(64, 64, 3, 0), (64, 3, 3, 0), (3, 64, 3, 0), (256, 64, 3, 2), (102, 128, 3, 0), (128, 102, 3, 0), (48, 160, 1, 0),
              (64, 320, 1, 0), (12, 3, 5, 0), (36, 64, 3, 3), (200, 72, 3, 0)]
    for dt in (torch.bfloat16, torch.float16):
        params = [(torch.nn.Parameter(torch.randn(co, ci, k, k, device="cuda") * 0.1), torch.nn.Parameter(torch.randn(co, device="cuda")), ps)
                  for (co, ci, k, ps) in shapes]
        grp = ops.PackGroup()
        with ops.forward_scope(grp):
            held = [(ops.pack_conv(w, b, dt, ps_r=ps), ops.pack_conv(w, None, dt, dgrad=True, ps_r=ps)) for (w, b, ps) in params]
        with torch.no_grad():
            for (w, b, _) in params:                              # an optimizer step: new values at the same addresses
                w.mul_(-0.7).add_(0.01)
                b.add_(0.5)
        with ops.forward_scope(grp):                              # the grouped launch refreshes every layout in place
            again = [(ops.pack_conv(w, b, dt, ps_r=ps), ops.pack_conv(w, None, dt, dgrad=True, ps_r=ps)) for (w, b, ps) in params]
        torch.cuda.synchronize()
        for (w, b, ps), (pf, pd), (qf, qd), shp in zip(params, held, again, shapes):
            assert qf.wpk.data_ptr() == pf.wpk.data_ptr() and qd.wpk.data_ptr() == pd.wpk.data_ptr(), "the group serves its own buffers"
            rf = ops.pack_conv(w.detach().clone().requires_grad_(False), b.detach().clone(), dt, ps_r=ps, cache=False)
            rd = ops.pack_conv(w.detach().clone(), None, dt, dgrad=True, ps_r=ps, cache=False)
            torch.cuda.synchronize()
            assert torch.equal(qf.wpk.view(torch.int16), rf.wpk.view(torch.int16)), ("forward", shp, dt)
            assert torch.equal(qf.bias, rf.bias), ("bias", shp, dt)
            assert torch.equal(qd.wpk.view(torch.int16), rd.wpk.view(torch.int16)), ("dgrad", shp, dt)
