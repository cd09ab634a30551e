"""numpy restatements of the leaf ops, independent of torch (oracle: test infrastructure).

Used to cross-check `torch.nn.functional` (and therefore oracle.functional) on
small shapes, and as the definition the HIP kernels are written against:

  conv2d_same   y[n,co,h,w] = b[co] + sum_{ci,kh,kw} w[co,ci,kh,kw] x[n,ci,h+kh-ph,w+kw-pw]   (zero pad)
                models/common.py:7-30 (nn.Conv2d, stride 1, padding k//2)
  pixel_shuffle out[n,c,h*r+i,w*r+j] = in[n,c*r*r+i*r+j,h,w]        models/common.py:133
  conv2d_same_grads  the three gradients of conv2d_same (dgrad / wgrad / bias grad)
"""
import numpy as np


def conv2d_same(x, w, b=None):
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    n, ci, H, W = x.shape
    co, ci2, kh, kw = w.shape
    assert ci == ci2
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((n, ci, H + 2 * ph, W + 2 * pw))
    xp[:, :, ph:ph + H, pw:pw + W] = x
    y = np.zeros((n, co, H, W))
    for i in range(kh):
        for j in range(kw):
            y += np.einsum("nchw,oc->nohw", xp[:, :, i:i + H, j:j + W], w[:, :, i, j])
    if b is not None:
        y += np.asarray(b, np.float64).reshape(1, co, 1, 1)
    return y


def conv2d_same_grads(x, w, dy):
    """Returns (dx, dw, db) for y = conv2d_same(x, w, b)."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    dy = np.asarray(dy, np.float64)
    n, ci, H, W = x.shape
    co, _, kh, kw = w.shape
    ph, pw = kh // 2, kw // 2
    xp = np.zeros((n, ci, H + 2 * ph, W + 2 * pw))
    xp[:, :, ph:ph + H, pw:pw + W] = x
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for i in range(kh):
        for j in range(kw):
            dw[:, :, i, j] = np.einsum("nohw,nchw->oc", dy, xp[:, :, i:i + H, j:j + W])
            dxp[:, :, i:i + H, j:j + W] += np.einsum("nohw,oc->nchw", dy, w[:, :, i, j])
    return dxp[:, :, ph:ph + H, pw:pw + W], dw, dy.sum(axis=(0, 2, 3))


def pixel_shuffle(x, r):
    n, c, h, w = x.shape
    assert c % (r * r) == 0
    co = c // (r * r)
    x = x.reshape(n, co, r, r, h, w)
    return x.transpose(0, 1, 4, 2, 5, 3).reshape(n, co, h * r, w * r)


def pixel_unshuffle(y, r):
    n, c, H, W = y.shape
    h, w = H // r, W // r
    y = y.reshape(n, c, h, r, w, r)
    return y.transpose(0, 1, 3, 5, 2, 4).reshape(n, c * r * r, h, w)


def global_avg_pool(x):
    return x.mean(axis=(2, 3), keepdims=True)


def channel_attention(x, w1, b1, w2, b2):
    """models/rcan.py:10-29 on numpy arrays; w1: (C/r, C), w2: (C, C/r)."""
    m = x.mean(axis=(2, 3))
    z = np.maximum(m @ w1.T + b1, 0.0)
    s = 1.0 / (1.0 + np.exp(-(z @ w2.T + b2)))
    return x * s[:, :, None, None]
