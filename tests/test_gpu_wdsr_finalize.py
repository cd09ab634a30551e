"""WDSR-B: every pointwise pair's weight gradient finalized by one launch."""


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


def test_wdsr_b_finalizes_all_pointwise_pairs_with_one_launch(A, monkeypatch):
    """WDSR-B's 16 blocks issue 16 pw_wgrad_finalize launches.  The grouped form (opt-in: it measured no faster, the
    slabs are cache-hot right behind the kernel that wrote them) -- every block's srk_pw_wgrad_partial fills its slabs and ONE
    srk_pw_wgrad_finalize_group sums them all when the pass's deferred gradients are flushed -- gives the same bits as a finalize per pair."""
    L = A._lib
    kw = dict(type="B", n_feats=128, n_resblocks=3, scale_factor=2)

    def grads(each):
        prev = A.ops._PW_FIN_EACH
        A.ops._PW_FIN_EACH = each
        try:
            torch.manual_seed(0)
            m = A.WDSR(precision="bf16", **kw).cuda()
            g = torch.Generator().manual_seed(3)
            lr, hr = torch.rand(2, 3, 24, 24, generator=g).cuda(), torch.rand(2, 3, 48, 48, generator=g).cuda()
            loss = m._calculate_losses(img_sr=m(lr), img_hr=hr)["loss"]
            loss.backward()
            torch.cuda.synchronize()
            return {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
        finally:
            A.ops._PW_FIN_EACH = prev

    calls = []
    real, real_check = L.call, L.check
    monkeypatch.setattr(L, "call", lambda name, *a, **k: (calls.append(name), real(name, *a, **k))[1])
    monkeypatch.setattr(L, "check", lambda rc, name="": (calls.append(name), real_check(rc, name))[1])
    ga = grads(False)
    assert calls.count("srk_pw_wgrad_partial") == 3 and calls.count("srk_pw_wgrad_finalize_group") == 1 and calls.count("srk_pw_wgrad") == 0, \
        [c for c in calls if "pw_wgrad" in c]
    calls.clear()
    gb = grads(True)
    assert calls.count("srk_pw_wgrad") == 3 and calls.count("srk_pw_wgrad_partial") == 0
    assert ga.keys() == gb.keys() and len(ga) > 10
    for k in ga:
        assert torch.equal(ga[k], gb[k]), k


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
