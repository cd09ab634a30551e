"""Cross-checks torch's leaf ops (what oracle.functional is built from) against the
independent numpy restatement in oracle/conv_np.py, incl. odd H/W, C=102 and N=1."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import conv_np


@pytest.mark.parametrize("n,ci,co,h,w,k", [(1, 3, 8, 5, 7, 3), (2, 16, 12, 6, 6, 3), (1, 102, 16, 4, 5, 3),
                                           (1, 8, 6, 5, 5, 1), (1, 3, 48, 6, 7, 5), (1, 3, 4, 12, 11, 9)])
def test_conv_and_grads(n, ci, co, h, w, k):
    rng = np.random.default_rng(1)
    x, wt, b = rng.standard_normal((n, ci, h, w)), rng.standard_normal((co, ci, k, k)) * 0.1, rng.standard_normal(co)
    dy = rng.standard_normal((n, co, h, w))
    xt = torch.tensor(x, requires_grad=True); wtt = torch.tensor(wt, requires_grad=True); bt = torch.tensor(b, requires_grad=True)
    y = F.conv2d(xt, wtt, bt, padding=k // 2)
    np.testing.assert_allclose(y.detach().numpy(), conv_np.conv2d_same(x, wt, b), atol=1e-10)
    y.backward(torch.tensor(dy))
    dx, dw, db = conv_np.conv2d_same_grads(x, wt, dy)
    np.testing.assert_allclose(xt.grad.numpy(), dx, atol=1e-10)
    np.testing.assert_allclose(wtt.grad.numpy(), dw, atol=1e-9)
    np.testing.assert_allclose(bt.grad.numpy(), db, atol=1e-10)


@pytest.mark.parametrize("r", [2, 3, 4])
def test_pixel_shuffle(r):
    x = np.arange(2 * 3 * r * r * 4 * 5, dtype=np.float64).reshape(2, 3 * r * r, 4, 5)
    ps = conv_np.pixel_shuffle(x, r)
    np.testing.assert_array_equal(F.pixel_shuffle(torch.tensor(x), r).numpy(), ps)
    np.testing.assert_array_equal(conv_np.pixel_unshuffle(ps, r), x)


def test_channel_attention():
    rng = np.random.default_rng(2)
    x = rng.standard_normal((2, 16, 5, 6)); w1 = rng.standard_normal((4, 16)); b1 = rng.standard_normal(4)
    w2 = rng.standard_normal((16, 4)); b2 = rng.standard_normal(16)
    from oracle import functional as OF
    sd = {"p.conv_du.0.weight": torch.tensor(w1).view(4, 16, 1, 1), "p.conv_du.0.bias": torch.tensor(b1),
          "p.conv_du.2.weight": torch.tensor(w2).view(16, 4, 1, 1), "p.conv_du.2.bias": torch.tensor(b2)}
    np.testing.assert_allclose(OF.ca_layer(sd, "p", torch.tensor(x)).numpy(),
                               conv_np.channel_attention(x, w1, b1, w2, b2), atol=1e-12)
