"""D-DBPN on the HIP path.  Reference: models/ddbpn.py:10-137 (same ctor, same state_dict keys).

The up / down projection units use nn.ConvTranspose2d / nn.Conv2d with kernel 6/8/12, stride 2/4/8, padding 2
(ddbpn.py:10-24).  Scale 4 (kernel 8, stride 4) in 16-bit runs on the DIRECT projection kernels (csrc/proj.hip through ops.proj_prelu /
ops.conv_general / ops.conv_transpose_general: no column tensor, the PReLU behind every projection fused into the launch); scales 2 and 8
and fp32 keep the NHWC im2col / col2im forms (strided conv = unfold + 1x1 MFMA conv, transposed conv = 1x1 MFMA conv + fold).  The
dense concatenations are channel slices of two buffers (ops.SliceBuffer: no copies, gradients accumulated in place); `b_0.sub(x)`
(ddbpn.py:57) stays a torch op on NHWC tensors."""
from typing import Any

import torch
import torch.nn as nn

from .. import ops
from .common import MeanShift
from .srmodel import SRModel


def projection_conv(in_channels, out_channels, scale, up=True):
    kernel_size, stride, padding = {2: (6, 2, 2), 4: (8, 4, 2), 8: (12, 8, 2)}[scale]
    conv_f = nn.ConvTranspose2d if up else nn.Conv2d
    return conv_f(in_channels, out_channels, kernel_size, stride=stride, padding=padding)


def _proj(seq, x):
    """[projection conv, PReLU] on NHWC x."""
    c, act = seq[0], seq[1]
    up = isinstance(c, nn.ConvTranspose2d)
    if act.weight.numel() in (1, 32) and ops.proj_ok(x, c.weight, c.stride[0], c.padding[0], up):
        return ops.proj_prelu(x, c.weight, c.bias, act.weight, up=up)      # scale 4, 16-bit: conv + PReLU in one launch (csrc/proj.hip)
    if up:
        y = ops.conv_transpose_general(x, c.weight, c.bias, stride=c.stride[0], pad=c.padding[0])
    else:
        y = ops.conv_general(x, c.weight, c.bias, stride=c.stride[0], pad=c.padding[0])
    return ops.prelu(y, act.weight)


class DenseProjection(nn.Module):
    """ddbpn.py:27-64"""

    def __init__(self, in_channels, nr, scale, up=True, bottleneck=True):
        super().__init__()
        if bottleneck:
            self.bottleneck = nn.Sequential(*[nn.Conv2d(in_channels, nr, 1), nn.PReLU(nr)])
            inter_channels = nr
        else:
            self.bottleneck = None
            inter_channels = in_channels
        self.conv_1 = nn.Sequential(*[projection_conv(inter_channels, nr, scale, up), nn.PReLU(nr)])
        self.conv_2 = nn.Sequential(*[projection_conv(nr, inter_channels, scale, not up), nn.PReLU(inter_channels)])
        self.conv_3 = nn.Sequential(*[projection_conv(inter_channels, nr, scale, up), nn.PReLU(nr)])

    def nhwc(self, x, dest=None):
        """`dest`: (ops.SliceBuffer, index) -- the slot of a concatenation buffer the unit's output is written to."""
        if self.bottleneck is not None:
            x = ops.prelu(ops.conv(x, self.bottleneck[0].weight, self.bottleneck[0].bias), self.bottleneck[1].weight)
        else:
            # x has TWO consumers here (conv_1 and `b_0.sub(x)`): autograd must sum their gradients itself.  A concatenation's
            # "add your data gradient into the shared buffer" marker (ops.concat_slices) is for a single consumer: a 1x1 conv behind it
            # (the transposed projection's GEMM at scales 2 / 8) would add in place AND be summed again.
            x.__dict__.pop("_srk_gacc", None)
        a_0 = _proj(self.conv_1, x)
        b_0 = _proj(self.conv_2, a_0)
        e = b_0.sub(x)
        a_1 = _proj(self.conv_3, e)
        return ops.add_into(a_0, a_1, dest)

    def forward(self, x):
        """NCHW float in / out, like the reference module."""
        dt = getattr(self, "compute_dtype", torch.float32)
        return ops.nhwc_to_nchw(self.nhwc(ops.nchw_to_nhwc(x, dt)), self.conv_1[0].out_channels)


class DDBPN(SRModel):
    def __init__(self, **kwargs: dict[str, Any]):
        super().__init__(**kwargs)
        n0, nr = 128, 32
        self.depth = 6
        if self._channels == 3:
            self.sub_mean = MeanShift()
        initial = [nn.Conv2d(self._channels, n0, 3, padding=1), nn.PReLU(n0), nn.Conv2d(n0, nr, 1), nn.PReLU(nr)]
        self.initial = nn.Sequential(*initial)
        self.upmodules = nn.ModuleList()
        self.downmodules = nn.ModuleList()
        channels = nr
        for i in range(self.depth):
            self.upmodules.append(DenseProjection(channels, nr, self._scale_factor, True, i > 1))
            if i != 0:
                channels += nr
        channels = nr
        for i in range(self.depth - 1):
            self.downmodules.append(DenseProjection(channels, nr, self._scale_factor, False, i != 0))
            channels += nr
        self.reconstruction = nn.Sequential(*[nn.Conv2d(self.depth * nr, self._channels, 3, padding=1)])
        if self._channels == 3:
            self.add_mean = MeanShift(sign=1)

    def forward(self, x):
        """ddbpn.py:112-137.  NCHW float in [0,1] -> NCHW fp32."""
        with ops.forward_scope(self._pack_group()):
            rgb = self._channels == 3
            i0, a0, i2, a2 = self.initial
            x = ops.prelu(ops.head_conv(x, i0.weight, i0.bias, self.sub_mean.neg_shift() if rgb else None, self.compute_dtype), a0.weight)
            x = ops.prelu(ops.conv(x, i2.weight, i2.bias), a2.weight)
            # the HR feature maps land in their slots of ONE buffer: the growing concatenations the down units and the reconstruction
            # read (ddbpn.py:116-134) are its prefixes -- no copy (they were 0.6 of 6.6 ms per step at the reference's batch)
            n, h, w, _ = x.shape
            r = self._scale_factor
            hbuf = ops.SliceBuffer(self.depth, accumulate_grads=True).alloc(n, h * r, w * r, x.shape[3], x.dtype, x.device)
            # ... and the LR feature maps of the down units likewise (`torch.cat(l_list)`, ddbpn.py:118,131: five copies forward and the
            # slice / add launches of their gradients, ~40 small torch launches per step at the reference's batch)
            lbuf = ops.SliceBuffer(self.depth - 1, accumulate_grads=True).alloc(n, h, w, x.shape[3], x.dtype, x.device)
            h_list, l_list = [], []
            for i in range(self.depth - 1):
                l = x if i == 0 else ops.concat_slices(lbuf, l_list)
                h_list.append(self.upmodules[i].nhwc(l, (hbuf, i)))
                l_list.append(self.downmodules[i].nhwc(ops.concat_slices(hbuf, h_list), (lbuf, i)))
            h_list.append(self.upmodules[-1].nhwc(ops.concat_slices(lbuf, l_list), (hbuf, self.depth - 1)))
            rec = self.reconstruction[0]
            return ops.tail_conv(ops.concat_slices(hbuf, h_list), rec.weight, rec.bias, post_add=self.add_mean.shift() if rgb else None)
