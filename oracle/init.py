"""State-dict construction in the reference's key layout and RNG order (oracle: test infrastructure).

`build_state_dict(cls, **ctor_kwargs)` returns (OrderedDict name -> tensor,
set of trainable names).  Parameters are drawn exactly as the reference's
constructors draw them -- `nn.Conv2d.reset_parameters` (kaiming-uniform a=sqrt(5),
bias U(+-1/sqrt(fan_in))) in the reference's construction ORDER, including the
throw-away draws of the two `MeanShift` convs (models/common.py:65) -- so that
`torch.manual_seed(s)` gives bit-identical weights (pinned by
tests/golden/manifest.json `init_seed0`).  Key layout: SURVEY.md 8(b).
"""
from collections import OrderedDict
from math import log2

import torch
import torch.nn as nn

from .functional import RDN_CONFIGS, RGB_MEAN


class StateDict(OrderedDict):
    """OrderedDict + `param_order`: the names `module.named_parameters()` yields, in order (first name of a shared
    parameter only, no buffers).  Equals the key order for every model without BatchNorm / shared modules."""
    param_order = None


class _Builder:
    def __init__(self):
        self.sd = StateDict()
        self.trainable = set()
        self.order = []

    def batch_norm(self, name, c, alias=None):
        """nn.BatchNorm2d: weight 1, bias 0, running_mean 0, running_var 1, num_batches_tracked 0 (no RNG draws).
        `alias`: a second name under which the reference registers the SAME module (common.py:97-98)."""
        w, b = torch.ones(c), torch.zeros(c)
        rm, rv, nb = torch.zeros(c), torch.ones(c), torch.tensor(0, dtype=torch.long)
        for nm in (name,):
            self._add(nm + ".weight", w)
            self._add(nm + ".bias", b)
            self.sd[nm + ".running_mean"], self.sd[nm + ".running_var"], self.sd[nm + ".num_batches_tracked"] = rm, rv, nb
        return (w, b, rm, rv, nb)

    def alias_batch_norm(self, name, tensors):
        w, b, rm, rv, nb = tensors                      # same tensor objects: shared parameters and buffers
        self.sd[name + ".weight"], self.sd[name + ".bias"] = w, b
        self.sd[name + ".running_mean"], self.sd[name + ".running_var"], self.sd[name + ".num_batches_tracked"] = rm, rv, nb

    def prelu(self, name, n=1):
        t = torch.full((n,), 0.25)                      # nn.PReLU(num_parameters=n, init=0.25)
        self._add(name + ".weight", t)
        return t

    def alias(self, name, t):
        self.sd[name] = t

    def conv_t(self, name, cin, cout, k, stride, pad):
        m = nn.ConvTranspose2d(cin, cout, k, stride=stride, padding=pad)      # consumes the RNG as the reference does
        self._add(name + ".weight", m.weight.detach().clone())
        self._add(name + ".bias", m.bias.detach().clone())

    def conv(self, name, cin, cout, k, weight_norm=False):
        m = nn.Conv2d(cin, cout, k)            # consumes the RNG as the reference does
        w, b = m.weight.detach().clone(), m.bias.detach().clone()
        if weight_norm:
            # legacy nn.utils.weight_norm: registers weight_g, weight_v after `bias`
            # (state_dict order: bias, weight_g, weight_v) -- models/wdsr.py:62
            g = w.flatten(1).norm(dim=1).view(-1, 1, 1, 1)
            self._add(name + ".bias", b)
            self._add(name + ".weight_g", g)
            self._add(name + ".weight_v", w)
        else:
            self._add(name + ".weight", w)
            self._add(name + ".bias", b)

    def mean_shift(self, name, sign):
        nn.Conv2d(3, 3, 1)                      # throw-away draw (models/common.py:65)
        self.sd[name + ".weight"] = torch.eye(3).view(3, 3, 1, 1).clone()
        self.sd[name + ".bias"] = sign * torch.tensor(RGB_MEAN)

    def _add(self, key, t):
        self.sd[key] = t
        self.trainable.add(key)
        self.order.append(key)


def _upscale(b, prefix, scale, n_feats):
    r = 2 if scale % 2 == 0 else 3
    for i in range(int(log2(scale))):
        b.conv(f"{prefix}.{2 * i}", n_feats, n_feats * r * r, 3)


def build_state_dict(cls, **kw):
    b = _Builder()
    scale = kw.get("scale_factor", 4)
    ch = kw.get("channels", 3)
    if cls == "EDSR":                               # models/edsr.py:13-38
        F_, B = kw.get("n_feats", 64), kw.get("n_resblocks", 16)
        if ch == 3:
            b.mean_shift("sub_mean", -1)
            b.mean_shift("add_mean", +1)
        b.conv("head.0", ch, F_, 3)
        for i in range(B):
            b.conv(f"body.{i}.body.0", F_, F_, 3)
            b.conv(f"body.{i}.body.2", F_, F_, 3)
        b.conv(f"body.{B}", F_, F_, 3)
        _upscale(b, "tail.0", scale, F_)
        b.conv("tail.1", F_, ch, 3)
    elif cls == "RCAN":                             # models/rcan.py:82-113
        F_, B, G, red = kw.get("n_feats", 64), kw.get("n_resblocks", 16), kw.get("n_resgroups", 10), kw.get("reduction", 16)
        if ch == 3:
            b.mean_shift("sub_mean", -1)
        b.conv("head.0", ch, F_, 3)
        for g in range(G):
            for r in range(B):
                p = f"body.{g}.body.{r}.body"
                b.conv(p + ".0", F_, F_, 3)
                b.conv(p + ".2", F_, F_, 3)
                b.conv(p + ".3.conv_du.0", F_, F_ // red, 1)
                b.conv(p + ".3.conv_du.2", F_ // red, F_, 1)
            b.conv(f"body.{g}.body.{B}", F_, F_, 3)
        b.conv(f"body.{G}", F_, F_, 3)
        _upscale(b, "tail.0", scale, F_)
        b.conv("tail.1", F_, ch, 3)
        if ch == 3:
            b.mean_shift("add_mean", +1)            # created last (rcan.py:112-113)
        # state_dict order follows attribute REGISTRATION order: sub_mean, head, body, tail, add_mean
        order = [k for k in b.sd if k.startswith("sub_mean")] + \
                [k for k in b.sd if k.startswith("head")] + [k for k in b.sd if k.startswith("body")] + \
                [k for k in b.sd if k.startswith("tail")] + [k for k in b.sd if k.startswith("add_mean")]
        b.sd = StateDict((k, b.sd[k]) for k in order)
    elif cls == "RDN":                              # models/rdn.py:47-97
        D, C, G = RDN_CONFIGS[kw.get("rdn_config", "B")]
        G0, k = kw.get("G0", 64), kw.get("kernel_size", 3)
        b.conv("SFENet1", ch, G0, k)
        b.conv("SFENet2", G0, G0, k)
        for d in range(D):
            for c in range(C):
                b.conv(f"_RDBs.{d}.convs.{c}.conv.0", G0 + c * G, G, 3)
            b.conv(f"_RDBs.{d}.LFF", G0 + C * G, G0, 1)
        b.conv("GFF.0", D * G0, G0, 1)
        b.conv("GFF.1", G0, G0, k)
        if scale in (2, 3):
            b.conv("UPNet.0", G0, G * scale * scale, k)
            b.conv("UPNet.2", G, 3, k)
        elif scale == 4:
            b.conv("UPNet.0", G0, G * 4, k)
            b.conv("UPNet.2", G, G * 4, k)
            b.conv("UPNet.4", G, ch, k)
        else:
            raise ValueError("scale must be 2 or 3 or 4.")
    elif cls == "WDSR":                             # models/wdsr.py:58-100
        F_, B, kind = kw.get("n_feats", 128), kw.get("n_resblocks", 16), kw.get("type", "B")
        out_feats = scale * scale * ch
        b.conv("head.0", ch, F_, 3, True)
        for i in range(B):
            if kind == "A":
                b.conv(f"body.{i}.body.0", F_, 4 * F_, 3, True)
                b.conv(f"body.{i}.body.2", 4 * F_, F_, 3, True)
            else:
                b.conv(f"body.{i}.body.0", F_, 6 * F_, 1, True)
                b.conv(f"body.{i}.body.2", 6 * F_, int(F_ * 0.8), 1, True)
                b.conv(f"body.{i}.body.3", int(F_ * 0.8), F_, 3, True)
        b.conv("tail.0", F_, out_feats, 3, True)
        b.conv("skip.0", 3, out_feats, 5, True)
    elif cls == "SRCNN":                            # models/srcnn.py:14-22
        b.conv("_net.0", ch, 64, 9)
        b.conv("_net.2", 64, 32, 1)
        b.conv("_net.4", 32, ch, 5)
    elif cls == "SRResNet":                         # models/srresnet.py:10-30
        F_, B = kw.get("n_feats", 64), kw.get("n_resblocks", 16)
        b.conv("head.0", ch, F_, 9)
        b.prelu("head.1")
        for i in range(B):
            p = f"body.{i}.body"
            # ctor argument order: norm = BatchNorm2d and act = PReLU are built BEFORE the block's convs (no RNG in either);
            # registration order inside the Sequential: conv, norm, act, conv, norm (the same instance again)
            b.conv(p + ".0", F_, F_, 3)
            bn = b.batch_norm(p + ".1", F_)
            b.prelu(p + ".2")
            b.conv(p + ".3", F_, F_, 3)
            b.alias_batch_norm(p + ".4", bn)
        b.conv(f"body.{B}.0", F_, F_, 3)
        b.batch_norm(f"body.{B}.1", F_)
        r = 2 if scale % 2 == 0 else 3
        act = None
        for i in range(int(log2(scale))):           # UpscaleBlock(act=PReLU): [conv, PixelShuffle, act] with ONE act instance
            b.conv(f"tail.0.{3 * i}", F_, F_ * r * r, 3)
            if act is None:
                act = b.prelu(f"tail.0.{3 * i + 2}")
            else:
                b.alias(f"tail.0.{3 * i + 2}.weight", act)
        b.conv("tail.1", F_, ch, 9)
    elif cls == "DDBPN":                            # models/ddbpn.py:71-110
        n0, nr, depth = 128, 32, 6
        k, st, pd = {2: (6, 2, 2), 4: (8, 4, 2), 8: (12, 8, 2)}[scale]

        def dense(prefix, cin, up, bottleneck):     # ddbpn.py:27-53
            if bottleneck:
                b.conv(prefix + ".bottleneck.0", cin, nr, 1)
                b.prelu(prefix + ".bottleneck.1", nr)
                inter = nr
            else:
                inter = cin
            for j, (ci, co, u) in enumerate(((inter, nr, up), (nr, inter, not up), (inter, nr, up)), start=1):
                if u:
                    b.conv_t(f"{prefix}.conv_{j}.0", ci, co, k, st, pd)
                else:
                    m = nn.Conv2d(ci, co, k, stride=st, padding=pd)
                    b._add(f"{prefix}.conv_{j}.0.weight", m.weight.detach().clone())
                    b._add(f"{prefix}.conv_{j}.0.bias", m.bias.detach().clone())
                b.prelu(f"{prefix}.conv_{j}.1", co)
        if ch == 3:
            b.mean_shift("sub_mean", -1)
        b.conv("initial.0", ch, n0, 3)
        b.prelu("initial.1", n0)
        b.conv("initial.2", n0, nr, 1)
        b.prelu("initial.3", nr)
        channels = nr
        for i in range(depth):
            dense(f"upmodules.{i}", channels, True, i > 1)
            if i != 0:
                channels += nr
        channels = nr
        for i in range(depth - 1):
            dense(f"downmodules.{i}", channels, False, i != 0)
            channels += nr
        b.conv("reconstruction.0", depth * nr, ch, 3)
        if ch == 3:
            b.mean_shift("add_mean", +1)
    else:
        raise KeyError(cls)
    b.sd.param_order = [k for k in b.order]
    return b.sd, b.trainable
