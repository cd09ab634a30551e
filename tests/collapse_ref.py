"""Float64 statement of the collapsed HR stage (csrc/hr_tail.hip, ops.HrTailFn) -- TEST INFRASTRUCTURE, never imported by the product.

The reference ends EDSR / RCAN / RDN with  conv3x3(Ci -> 4C) -> nn.PixelShuffle(2) -> conv3x3(C -> O)  (models/common.py:112-139,
edsr.py:48-52, rcan.py tail, rdn.py:85-95) and no activation in between, so the image is ONE linear map of the upsampler's input.
`layerwise` is the reference's form (plain torch ops); `collapse` / `collapsed_forward` build the 5x5 convolution and its border
terms from the two layers' weights; `np_border_sums` / `np_expand` are the explicit chain-rule formulas the HIP kernels implement
(`srk_hrtail_edge_bwd_w`, `srk_hrtail_expand`).  tests/test_collapse_math.py checks all of them against autograd of `layerwise`,
tests/test_gpu_hr_tail.py checks the kernels against them."""
import numpy as np
import torch
import torch.nn.functional as F

def sub(a, d):      # a in {0,1}, d in {-1,0,1} -> (i, s)
    ap = a + d
    return ap % 2, ap // 2

def collapse(Wt, bt, Wu, bu):
    """-> Weff [O*4, Ci, 5, 5], beff [O*4]; edge corrections dict name -> (Wc, bc) (full 12-channel, zeros where not applicable)"""
    O_, C_ = Wt.shape[:2]; Ci_ = Wu.shape[1]
    Wu5 = Wu.view(C_, 2, 2, Ci_, 3, 3)        # [c][i][j][ci][ey][ex]
    bu3 = bu.view(C_, 2, 2)
    def build(pred):
        W = torch.zeros(O_, 2, 2, Ci_, 5, 5, dtype=Wt.dtype); b = torch.zeros(O_, 2, 2, dtype=Wt.dtype)
        for a in range(2):
            for bb in range(2):
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        if not pred(a, bb, dy, dx): continue
                        i, sy = sub(a, dy); j, sx = sub(bb, dx)
                        wt = Wt[:, :, dy + 1, dx + 1]                 # [o][c]
                        b[:, a, bb] += wt @ bu3[:, i, j]
                        for ey in (-1, 0, 1):
                            for ex in (-1, 0, 1):
                                W[:, a, bb, :, sy + ey + 2, sx + ex + 2] += wt @ Wu5[:, i, j, :, ey + 1, ex + 1]
        return W.reshape(O_ * 4, Ci_, 5, 5), b.reshape(O_ * 4)
    full = build(lambda a, b, dy, dx: True)
    full = (full[0], full[1] + bt.repeat_interleave(4))
    top = build(lambda a, b, dy, dx: a == 0 and dy == -1)
    bot = build(lambda a, b, dy, dx: a == 1 and dy == 1)
    left = build(lambda a, b, dy, dx: b == 0 and dx == -1)
    right = build(lambda a, b, dy, dx: b == 1 and dx == 1)
    tl = build(lambda a, b, dy, dx: a == 0 and dy == -1 and b == 0 and dx == -1)
    tr = build(lambda a, b, dy, dx: a == 0 and dy == -1 and b == 1 and dx == 1)
    bl = build(lambda a, b, dy, dx: a == 1 and dy == 1 and b == 0 and dx == -1)
    br = build(lambda a, b, dy, dx: a == 1 and dy == 1 and b == 1 and dx == 1)
    return full, dict(top=top, bot=bot, left=left, right=right, tl=tl, tr=tr, bl=bl, br=br)

def collapsed_forward(X, Wt, bt, Wu, bu):
    (We, be), cor = collapse(Wt, bt, Wu, bu)
    T = F.conv2d(X, We, be, padding=2)                      # [N, 12, H, W]
    H, W = X.shape[2:]
    cv = lambda k: F.conv2d(X, cor[k][0], cor[k][1], padding=2)
    T = T.clone()
    T[:, :, 0, :] -= cv('top')[:, :, 0, :]
    T[:, :, H - 1, :] -= cv('bot')[:, :, H - 1, :]
    T[:, :, :, 0] -= cv('left')[:, :, :, 0]
    T[:, :, :, W - 1] -= cv('right')[:, :, :, W - 1]
    T[:, :, 0, 0] += cv('tl')[:, :, 0, 0]
    T[:, :, 0, W - 1] += cv('tr')[:, :, 0, W - 1]
    T[:, :, H - 1, 0] += cv('bl')[:, :, H - 1, 0]
    T[:, :, H - 1, W - 1] += cv('br')[:, :, H - 1, W - 1]
    return F.pixel_shuffle(T, 2)

def layerwise(X, Wt, bt, Wu, bu):
    return F.conv2d(F.pixel_shuffle(F.conv2d(X, Wu, bu, padding=1), 2), Wt, bt, padding=1)



# ---------------- explicit formulas (what the HIP kernels implement) ----------------

def np_border_sums(X, g):
    """X [N,Ci,H,W], g [N,O,2H,2W] (numpy) -> R [4O,Ci,5,5], r0 [4O], E [4,2O,Ci,5], e0 [4,2O], K [4,O,Ci], k0 [4,O]"""
    N, Ci, H, W = X.shape; O = g.shape[1]
    Xp = np.pad(X, ((0, 0), (0, 0), (2, 2), (2, 2)))
    g12 = np.zeros((N, O, 2, 2, H, W))
    for a in range(2):
        for b in range(2):
            g12[:, :, a, b] = g[:, :, a::2, b::2]
    R = np.zeros((O, 2, 2, Ci, 5, 5)); r0 = g12.sum(axis=(0, 4, 5))
    for fy in range(5):
        for fx in range(5):
            R[:, :, :, :, fy, fx] = np.einsum('noabyx,ncyx->oabc', g12, Xp[:, :, fy:fy + H, fx:fx + W])
    E = np.zeros((4, O, 2, Ci, 5)); e0 = np.zeros((4, O, 2)); K = np.zeros((4, O, Ci)); k0 = np.zeros((4, O))
    for t in range(5):
        # top: a = 0, row 0 ; kk = (o, b)
        E[0, :, :, :, t] = np.einsum('nobx,ncx->obc', g12[:, :, 0, :, 0, :], Xp[:, :, 2, t:t + W])
        E[1, :, :, :, t] = np.einsum('nobx,ncx->obc', g12[:, :, 1, :, H - 1, :], Xp[:, :, 2 + H - 1, t:t + W])
        # left: b = 0, col 0 ; kk = (o, a)
        E[2, :, :, :, t] = np.einsum('noay,ncy->oac', g12[:, :, :, 0, :, 0], Xp[:, :, t:t + H, 2])
        E[3, :, :, :, t] = np.einsum('noay,ncy->oac', g12[:, :, :, 1, :, W - 1], Xp[:, :, t:t + H, 2 + W - 1])
    e0[0] = g12[:, :, 0, :, 0, :].sum(axis=(0, 3)); e0[1] = g12[:, :, 1, :, H - 1, :].sum(axis=(0, 3))
    e0[2] = g12[:, :, :, 0, :, 0].sum(axis=(0, 3)); e0[3] = g12[:, :, :, 1, :, W - 1].sum(axis=(0, 3))
    cor = [(0, 0, 0, 0), (0, 1, 0, W - 1), (1, 0, H - 1, 0), (1, 1, H - 1, W - 1)]     # (a, b, y, x)
    for c, (a, b, y, x) in enumerate(cor):
        K[c] = np.einsum('no,nc->oc', g12[:, :, a, b, y, x], X[:, :, y, x]); k0[c] = g12[:, :, a, b, y, x].sum(axis=0)
    return R.reshape(O * 4, Ci, 5, 5), r0.reshape(O * 4), E.reshape(4, O * 2, Ci, 5), e0.reshape(4, O * 2), K, k0

def np_expand(R, r0, E, e0, K, k0, Wt, Wu, bu):
    O, C = Wt.shape[:2]; Ci = Wu.shape[1]
    dWt = np.zeros_like(Wt); dbt = np.zeros(O); dWu = np.zeros_like(Wu); dbu = np.zeros(4 * C)
    for o in range(O):
        dbt[o] = sum(r0[o * 4 + a * 2 + b] for a in range(2) for b in range(2))
    def G(o, a, b, dy, dx, ci, fy, fx):
        v = R[o * 4 + a * 2 + b, ci, fy + 2, fx + 2]
        rowt = 0 if (a == 0 and dy == -1) else (1 if (a == 1 and dy == 1) else -1)
        colt = 2 if (b == 0 and dx == -1) else (3 if (b == 1 and dx == 1) else -1)
        if rowt >= 0 and fy == 0: v = v - E[rowt, o * 2 + b, ci, fx + 2]
        if colt >= 0 and fx == 0: v = v - E[colt, o * 2 + a, ci, fy + 2]
        if rowt >= 0 and colt >= 0 and fy == 0 and fx == 0: v = v + K[a * 2 + b, o, ci]
        return v
    def Gb(o, a, b, dy, dx):
        v = r0[o * 4 + a * 2 + b]
        rowt = 0 if (a == 0 and dy == -1) else (1 if (a == 1 and dy == 1) else -1)
        colt = 2 if (b == 0 and dx == -1) else (3 if (b == 1 and dx == 1) else -1)
        if rowt >= 0: v = v - e0[rowt, o * 2 + b]
        if colt >= 0: v = v - e0[colt, o * 2 + a]
        if rowt >= 0 and colt >= 0: v = v + k0[a * 2 + b, o]
        return v
    for a in range(2):
        for b in range(2):
            for dy in (-1, 0, 1):
                for dx in (-1, 0, 1):
                    i, sy = sub(a, dy); j, sx = sub(b, dx)
                    for o in range(O):
                        gb = Gb(o, a, b, dy, dx)
                        for c in range(C):
                            cu = c * 4 + i * 2 + j
                            dbu[cu] += Wt[o, c, dy + 1, dx + 1] * gb
                            dWt[o, c, dy + 1, dx + 1] += gb * bu[cu]
                            for ci in range(Ci):
                                for ey in (-1, 0, 1):
                                    for ex in (-1, 0, 1):
                                        gv = G(o, a, b, dy, dx, ci, sy + ey, sx + ex)
                                        dWu[cu, ci, ey + 1, ex + 1] += Wt[o, c, dy + 1, dx + 1] * gv
                                        dWt[o, c, dy + 1, dx + 1] += gv * Wu[cu, ci, ey + 1, ex + 1]
    return dWt, dbt, dWu, dbu
