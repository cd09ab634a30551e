"""Functional CPU restatement of the reference forward passes (oracle: test infrastructure).

Each function takes `sd`, a dict {reference state_dict key -> tensor}, and the
input tensor (NCHW float, as the reference: SURVEY.md 8(b)), and returns what
the reference's module of that name returns.  Autograd works through them, so
the backward oracle is torch's own differentiation of these expressions.

All file:line citations are relative to /root/reference.
"""
from math import log2

import torch
import torch.nn.functional as F

RGB_MEAN = (0.4488, 0.4371, 0.4040)   # models/common.py:62, models/wdsr.py:66-67


def conv_same(sd, prefix, x):
    """`DefaultConv2d` / `nn.Conv2d(..., padding=k//2)`: stride 1, zero pad k//2, bias.

    models/common.py:7-30; models/rdn.py:15,57-60; models/wdsr.py:17-20."""
    w = sd[prefix + ".weight"]
    b = sd.get(prefix + ".bias")
    return F.conv2d(x, w, b, stride=1, padding=(w.shape[2] // 2, w.shape[3] // 2))


def mean_shift(sd, prefix, x):
    """`MeanShift`: frozen 1x1 conv, W = I/std, b = sign*range*mean/std (models/common.py:58-71)."""
    return F.conv2d(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def res_block(sd, prefix, x, res_scale):
    """`ResBlock.forward` with n_conv_layers=2, no norm: conv -> ReLU -> conv, *res_scale, += x.

    models/common.py:74-109 (body indices 0 and 2, ReLU at 1)."""
    r = conv_same(sd, prefix + ".body.0", x)
    r = F.relu(r)
    r = conv_same(sd, prefix + ".body.2", r)
    r = r * res_scale
    return r + x


def upscale_block(sd, prefix, x, scale, n_feats):
    """`UpscaleBlock`: int(log2(s)) x [conv3x3(F -> F r^2), PixelShuffle(r)], r = 2 if s even else 3.

    models/common.py:112-139.  (For s=3 that is ONE stage since int(log2(3)) == 1.)"""
    assert scale in (2, 3, 4, 8)
    r = 2 if scale % 2 == 0 else 3
    for i in range(int(log2(scale))):
        x = conv_same(sd, f"{prefix}.{2 * i}", x)
        x = F.pixel_shuffle(x, r)
    return x


def edsr_forward(sd, x, *, n_resblocks, res_scale, scale, channels=3, n_feats=None):
    """`EDSR.forward` (models/edsr.py:40-54; ctor :13-38)."""
    if channels == 3:
        x = mean_shift(sd, "sub_mean", x)
    x = conv_same(sd, "head.0", x)
    res = x
    for i in range(n_resblocks):
        res = res_block(sd, f"body.{i}", res, res_scale)
    res = conv_same(sd, f"body.{n_resblocks}", res)
    res = res + x
    y = upscale_block(sd, "tail.0", res, scale, n_feats)
    y = conv_same(sd, "tail.1", y)
    if channels == 3:
        y = mean_shift(sd, "add_mean", y)
    return y


def ca_layer(sd, prefix, x):
    """`CALayer.forward`: GAP -> 1x1 (C -> C/r) -> ReLU -> 1x1 (C/r -> C) -> sigmoid -> x*y.

    models/rcan.py:10-29."""
    y = x.mean(dim=(2, 3), keepdim=True)                     # AdaptiveAvgPool2d(1)
    y = F.conv2d(y, sd[prefix + ".conv_du.0.weight"], sd[prefix + ".conv_du.0.bias"])
    y = F.relu(y)
    y = F.conv2d(y, sd[prefix + ".conv_du.2.weight"], sd[prefix + ".conv_du.2.bias"])
    return x * torch.sigmoid(y)


def rcab(sd, prefix, x):
    """`RCAB.forward`: conv -> ReLU -> conv -> CA, += x; res_scale NOT applied (models/rcan.py:33-55)."""
    r = conv_same(sd, prefix + ".body.0", x)
    r = F.relu(r)
    r = conv_same(sd, prefix + ".body.2", r)
    r = ca_layer(sd, prefix + ".body.3", r)
    return r + x


def residual_group(sd, prefix, x, n_resblocks):
    """`ResidualGroup.forward`: n x RCAB, conv, += x (models/rcan.py:59-74)."""
    r = x
    for b in range(n_resblocks):
        r = rcab(sd, f"{prefix}.body.{b}", r)
    r = conv_same(sd, f"{prefix}.body.{n_resblocks}", r)
    return r + x


def rcan_forward(sd, x, *, n_resgroups, n_resblocks, scale, channels=3, n_feats=None):
    """`RCAN.forward` (models/rcan.py:115-129; ctor :82-113)."""
    if channels == 3:
        x = mean_shift(sd, "sub_mean", x)
    x = conv_same(sd, "head.0", x)
    res = x
    for g in range(n_resgroups):
        res = residual_group(sd, f"body.{g}", res, n_resblocks)
    res = conv_same(sd, f"body.{n_resgroups}", res)
    res = res + x
    y = upscale_block(sd, "tail.0", res, scale, n_feats)
    y = conv_same(sd, "tail.1", y)
    if channels == 3:
        y = mean_shift(sd, "add_mean", y)
    return y


RDN_CONFIGS = {"A": (20, 6, 32), "B": (16, 8, 64)}   # models/rdn.py:51-54  (D, C, G)


def rdb(sd, prefix, x, n_conv):
    """`_RDB.forward`: C x [relu(conv3x3(cat)) appended to cat], LFF 1x1, + x (models/rdn.py:9-40)."""
    feat = x
    for c in range(n_conv):
        out = F.relu(conv_same(sd, f"{prefix}.convs.{c}.conv.0", feat))
        feat = torch.cat((feat, out), 1)
    return conv_same(sd, prefix + ".LFF", feat) + x


def rdn_forward(sd, x, *, rdn_config, scale):
    """`RDN.forward` (models/rdn.py:99-111; ctor :47-97).  No MeanShift."""
    D, C, _G = RDN_CONFIGS[rdn_config]
    f1 = conv_same(sd, "SFENet1", x)
    x = conv_same(sd, "SFENet2", f1)
    outs = []
    for d in range(D):
        x = rdb(sd, f"_RDBs.{d}", x, C)
        outs.append(x)
    x = conv_same(sd, "GFF.0", torch.cat(outs, 1))
    x = conv_same(sd, "GFF.1", x)
    x = x + f1
    if scale in (2, 3):                                     # rdn.py:76-83
        x = conv_same(sd, "UPNet.0", x)
        x = F.pixel_shuffle(x, scale)
        return conv_same(sd, "UPNet.2", x)
    if scale == 4:                                          # rdn.py:84-95
        x = F.pixel_shuffle(conv_same(sd, "UPNet.0", x), 2)
        x = F.pixel_shuffle(conv_same(sd, "UPNet.2", x), 2)
        return conv_same(sd, "UPNet.4", x)
    raise ValueError("scale must be 2 or 3 or 4.")


def wn_weight(sd, prefix):
    """Legacy `nn.utils.weight_norm` (dim=0): w = g * v / ||v||_2 over dims (1,2,3) (models/wdsr.py:62)."""
    v = sd[prefix + ".weight_v"]
    g = sd[prefix + ".weight_g"]
    norm = v.flatten(1).norm(dim=1).view(-1, 1, 1, 1)
    return v * (g / norm)


def wn_conv(sd, prefix, x):
    w = wn_weight(sd, prefix)
    return F.conv2d(x, w, sd[prefix + ".bias"], padding=(w.shape[2] // 2, w.shape[3] // 2))


def wdsr_block(sd, prefix, x, kind, res_scale):
    """`_Block_A` (models/wdsr.py:9-27) / `_Block_B` (:30-51): body, *res_scale, += x."""
    if kind == "A":
        r = wn_conv(sd, prefix + ".body.0", x)
        r = F.relu(r)
        r = wn_conv(sd, prefix + ".body.2", r)
    else:
        r = wn_conv(sd, prefix + ".body.0", x)       # 1x1 F -> 6F
        r = F.relu(r)
        r = wn_conv(sd, prefix + ".body.2", r)       # 1x1 6F -> int(0.8F)
        r = wn_conv(sd, prefix + ".body.3", r)       # 3x3 -> F
    return r * res_scale + x


def wdsr_forward(sd, x, *, kind, n_resblocks, res_scale, scale, channels=3):
    """`WDSR.forward` (models/wdsr.py:102-117; ctor :58-100)."""
    mean = None
    if channels == 3:
        mean = torch.tensor(RGB_MEAN, dtype=x.dtype, device=x.device).view(1, 3, 1, 1)
        x = x - mean
    s = F.pixel_shuffle(wn_conv(sd, "skip.0", x), scale)
    x = wn_conv(sd, "head.0", x)
    for i in range(n_resblocks):
        x = wdsr_block(sd, f"body.{i}", x, kind, res_scale)
    x = F.pixel_shuffle(wn_conv(sd, "tail.0", x), scale)
    x = x + s
    if channels == 3:
        x = x + mean
    return x


def srcnn_forward(sd, x, *, scale):
    """`SRCNN.forward`: bicubic x s, 9x9 -> ReLU -> 1x1 -> ReLU -> 5x5 (models/srcnn.py:9-27)."""
    x = F.interpolate(x, scale_factor=scale, mode="bicubic")
    x = F.relu(conv_same(sd, "_net.0", x))
    x = F.relu(conv_same(sd, "_net.2", x))
    return conv_same(sd, "_net.4", x)


def batch_norm(sd, prefix, x, training=True, momentum=0.1, eps=1e-5):
    """`nn.BatchNorm2d` (models/srresnet.py:17,20 via common.py:97-98): batch statistics + running-buffer update in
    training mode (the state dict's `running_mean` / `running_var` / `num_batches_tracked` are updated in place, like
    the module's buffers), running statistics in eval mode."""
    if training:
        sd[prefix + ".num_batches_tracked"] += 1
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training, momentum, eps)


def prelu(sd, prefix, x):
    """`nn.PReLU` with one shared slope (srresnet.py:14,20,27) or one per channel (ddbpn.py:33,42-53,82-86)."""
    return F.prelu(x, sd[prefix + ".weight"])


def res_block_bn(sd, prefix, x, res_scale, training=True):
    """`ResBlock.forward` with norm = BatchNorm2d and act = PReLU (models/common.py:74-109 as built by srresnet.py:16-19):
    body = [conv, bn, prelu, conv, bn] where indices 1 and 4 are ONE BatchNorm2d instance (same tensors, both keys)."""
    r = conv_same(sd, prefix + ".body.0", x)
    r = batch_norm(sd, prefix + ".body.1", r, training)
    r = prelu(sd, prefix + ".body.2", r)
    r = conv_same(sd, prefix + ".body.3", r)
    r = batch_norm(sd, prefix + ".body.4", r, training)
    return r * res_scale + x


def srresnet_forward(sd, x, *, n_resblocks, scale, training=True):
    """`SRResNet.forward` (models/srresnet.py:32-36; ctor :10-30): head = BasicBlock(9x9 conv, PReLU); body =
    n x ResBlock(BN, PReLU) + BasicBlock(3x3 conv, BN), `+ x`; tail = UpscaleBlock(act = one shared PReLU) + 9x9 conv."""
    x = prelu(sd, "head.1", conv_same(sd, "head.0", x))
    r = x
    for i in range(n_resblocks):
        r = res_block_bn(sd, f"body.{i}", r, 1.0, training)
    r = batch_norm(sd, f"body.{n_resblocks}.1", conv_same(sd, f"body.{n_resblocks}.0", r), training)
    x = r + x
    rr = 2 if scale % 2 == 0 else 3
    for i in range(int(log2(scale))):                        # UpscaleBlock with act: [conv, PixelShuffle, act] per stage
        x = F.pixel_shuffle(conv_same(sd, f"tail.0.{3 * i}", x), rr)
        x = prelu(sd, f"tail.0.{3 * i + 2}", x)
    return conv_same(sd, "tail.1", x)


DDBPN_PROJ = {2: (6, 2, 2), 4: (8, 4, 2), 8: (12, 8, 2)}     # models/ddbpn.py:11-15 (kernel, stride, padding)


def projection(sd, prefix, x, scale, up):
    """`projection_conv` + PReLU (models/ddbpn.py:10-24,40-53): ConvTranspose2d when `up`, else Conv2d."""
    _, stride, pad = DDBPN_PROJ[scale]
    w, b = sd[prefix + ".0.weight"], sd[prefix + ".0.bias"]
    y = F.conv_transpose2d(x, w, b, stride=stride, padding=pad) if up else F.conv2d(x, w, b, stride=stride, padding=pad)
    return prelu(sd, prefix + ".1", y)


def dense_projection(sd, prefix, x, scale, up, bottleneck):
    """`DenseProjection.forward` (models/ddbpn.py:55-64)."""
    if bottleneck:
        x = prelu(sd, prefix + ".bottleneck.1", F.conv2d(x, sd[prefix + ".bottleneck.0.weight"], sd[prefix + ".bottleneck.0.bias"]))
    a_0 = projection(sd, prefix + ".conv_1", x, scale, up)
    b_0 = projection(sd, prefix + ".conv_2", a_0, scale, not up)
    e = b_0.sub(x)
    a_1 = projection(sd, prefix + ".conv_3", e, scale, up)
    return a_0.add(a_1)


def ddbpn_forward(sd, x, *, scale, channels=3, depth=6):
    """`DDBPN.forward` (models/ddbpn.py:112-137; ctor :71-110)."""
    if channels == 3:
        x = mean_shift(sd, "sub_mean", x)
    x = prelu(sd, "initial.1", conv_same(sd, "initial.0", x))
    x = prelu(sd, "initial.3", conv_same(sd, "initial.2", x))
    h_list, l_list = [], []
    for i in range(depth - 1):
        l = x if i == 0 else torch.cat(l_list, dim=1)
        h_list.append(dense_projection(sd, f"upmodules.{i}", l, scale, True, i > 1))
        l_list.append(dense_projection(sd, f"downmodules.{i}", torch.cat(h_list, dim=1), scale, False, i != 0))
    h_list.append(dense_projection(sd, f"upmodules.{depth - 1}", torch.cat(l_list, dim=1), scale, True, depth - 1 > 1))
    out = conv_same(sd, "reconstruction.0", torch.cat(h_list, dim=1))
    if channels == 3:
        out = mean_shift(sd, "add_mean", out)
    return out


def forward(cls, sd, x, **kw):
    """Dispatch on the reference class name with the reference's ctor kwargs."""
    scale = kw.get("scale_factor", 4)
    channels = kw.get("channels", 3)
    if cls == "EDSR":
        return edsr_forward(sd, x, n_resblocks=kw.get("n_resblocks", 16), res_scale=kw.get("res_scale", 1),
                            scale=scale, channels=channels, n_feats=kw.get("n_feats", 64))
    if cls == "RCAN":
        return rcan_forward(sd, x, n_resgroups=kw.get("n_resgroups", 10), n_resblocks=kw.get("n_resblocks", 16),
                            scale=scale, channels=channels, n_feats=kw.get("n_feats", 64))
    if cls == "RDN":
        return rdn_forward(sd, x, rdn_config=kw.get("rdn_config", "B"), scale=scale)
    if cls == "WDSR":
        return wdsr_forward(sd, x, kind=kw.get("type", "B"), n_resblocks=kw.get("n_resblocks", 16),
                            res_scale=kw.get("res_scale", 1), scale=scale, channels=channels)
    if cls == "SRCNN":
        return srcnn_forward(sd, x, scale=scale)
    if cls == "SRResNet":
        return srresnet_forward(sd, x, n_resblocks=kw.get("n_resblocks", 16), scale=scale, training=kw.get("training", True))
    if cls == "DDBPN":
        return ddbpn_forward(sd, x, scale=scale, channels=channels)
    raise KeyError(cls)
