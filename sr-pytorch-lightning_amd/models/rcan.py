"""RCAN on the HIP path.  Reference: models/rcan.py:10-129 (same ctor, same state_dict keys)."""
from typing import Any

import torch.nn as nn

from .. import ops
from .common import DefaultConv2d, MeanShift, UpscaleBlock, _NCHWContract, upscale_tail
from .srmodel import SRModel


class CALayer(nn.Module):
    """Channel attention (rcan.py:10-29).  Parameters only; the arithmetic is fused into RCAB's kernels."""

    def __init__(self, channel, reduction=16):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.conv_du = nn.Sequential(
            nn.Conv2d(channel, channel // reduction, 1, padding=0, bias=True),
            nn.ReLU(inplace=True),
            nn.Conv2d(channel // reduction, channel, 1, padding=0, bias=True),
            nn.Sigmoid())


class RCAB(_NCHWContract, nn.Module):
    """conv -> ReLU -> conv -> CA, += x; `res_scale` is stored but NOT applied (rcan.py:33-55)."""

    def __init__(self, conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=nn.ReLU(True), res_scale=1):
        super().__init__()
        if bn:
            raise NotImplementedError("BatchNorm is not on the hot path")
        modules_body = []
        for i in range(2):
            modules_body.append(conv(in_channels=n_feat, out_channels=n_feat, kernel_size=kernel_size, bias=bias))
            if i == 0:
                modules_body.append(act)
        modules_body.append(CALayer(n_feat, reduction))
        self.body = nn.Sequential(*modules_body)
        self.res_scale = res_scale

    def _cout(self):
        return self.body[0].out_channels

    def nhwc(self, x):
        c1, c2, ca = self.body[0], self.body[2], self.body[3]
        return ops.rcab(x, c1.weight, c1.bias, c2.weight, c2.bias,
                        ca.conv_du[0].weight, ca.conv_du[0].bias, ca.conv_du[2].weight, ca.conv_du[2].bias)


class ResidualGroup(_NCHWContract, nn.Module):
    """n x RCAB, conv, += x (rcan.py:59-74)."""

    def __init__(self, conv, n_feat, kernel_size, reduction, act, res_scale, n_resblocks):
        super().__init__()
        modules_body = [RCAB(conv, n_feat, kernel_size, reduction, bias=True, bn=False, act=nn.ReLU(True), res_scale=1)
                        for _ in range(n_resblocks)]
        modules_body.append(conv(in_channels=n_feat, out_channels=n_feat, kernel_size=kernel_size))
        self.body = nn.Sequential(*modules_body)

    def _cout(self):
        return self.body[-1].out_channels

    def nhwc(self, x):
        blocks = []
        for blk in list(self.body)[:-1]:
            c1, c2, ca = blk.body[0], blk.body[2], blk.body[3]
            blocks.append((c1.weight, c1.bias, c2.weight, c2.bias,
                           ca.conv_du[0].weight, ca.conv_du[0].bias, ca.conv_du[2].weight, ca.conv_du[2].bias))
        r = ops.rcab_chain(x, blocks)
        return self.body[-1].nhwc(r, res=x)


class RCAN(SRModel):
    def __init__(self, n_feats: int = 64, n_resblocks: int = 16, n_resgroups: int = 10, reduction: int = 16,
                 res_scale: int = 1, **kwargs: dict[str, Any]):
        super().__init__(**kwargs)
        kernel_size = 3
        if self._channels == 3:
            self.sub_mean = MeanShift()
        modules_head = [DefaultConv2d(in_channels=self._channels, out_channels=n_feats, kernel_size=kernel_size)]
        modules_body = [ResidualGroup(DefaultConv2d, n_feats, kernel_size, reduction, act=nn.ReLU(True),
                                      res_scale=res_scale, n_resblocks=n_resblocks) for _ in range(n_resgroups)]
        modules_body.append(DefaultConv2d(in_channels=n_feats, out_channels=n_feats, kernel_size=kernel_size))
        modules_tail = [UpscaleBlock(self._scale_factor, n_feats),
                        DefaultConv2d(in_channels=n_feats, out_channels=self._channels, kernel_size=kernel_size)]
        self.head = nn.Sequential(*modules_head)
        self.body = nn.Sequential(*modules_body)
        self.tail = nn.Sequential(*modules_tail)
        if self._channels == 3:
            self.add_mean = MeanShift(sign=1)

    def body_nhwc(self, f):
        """The residual-in-residual trunk on NHWC features (rcan.py:119-122); see EDSR.body_nhwc."""
        r = f
        for grp in list(self.body)[:-1]:
            r = ops.cut(grp.nhwc(r))
        return self.body[-1].nhwc(r, res=f)

    def body_conv_launches(self):
        """(3x3 F -> F convolutions of the trunk per forward pass, F)"""
        groups = list(self.body)[:-1]
        return sum(2 * (len(g.body) - 1) + 1 for g in groups) + 1, self.body[-1].out_channels

    def forward(self, x):
        """rcan.py:115-129"""
        with ops.forward_scope(self._pack_group()):
            rgb = self._channels == 3
            f = ops.head_conv(x, self.head[0].weight, self.head[0].bias, self.sub_mean.neg_shift() if rgb else None,
                              self.compute_dtype)
            f = ops.cut(f, keep=True)
            r = self.body_nhwc(f)
            up = list(self.tail[0])
            return upscale_tail(r, [(c, p.upscale_factor) for c, p in zip(up[0::2], up[1::2])], self.tail[1],
                                post_add=self.add_mean.shift() if rgb else None)
