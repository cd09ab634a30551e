"""EDSR on the HIP path.  Reference: models/edsr.py:9-54 (same ctor, same state_dict keys)."""
from typing import Any

import torch.nn as nn

from .. import ops
from .common import DefaultConv2d, MeanShift, ResBlock, UpscaleBlock, upscale_tail
from .srmodel import SRModel


class EDSR(SRModel):
    def __init__(self, n_feats: int = 64, n_resblocks: int = 16, res_scale: int = 1, **kwargs: dict[str, Any]):
        super().__init__(**kwargs)
        kernel_size = 3
        if self._channels == 3:
            self.sub_mean = MeanShift()
            self.add_mean = MeanShift(sign=1)
        m_head = [DefaultConv2d(in_channels=self._channels, out_channels=n_feats, kernel_size=kernel_size)]
        m_body = [ResBlock(n_feats=n_feats, kernel_size=kernel_size, res_scale=res_scale) for _ in range(n_resblocks)]
        m_body.append(DefaultConv2d(in_channels=n_feats, out_channels=n_feats, kernel_size=kernel_size))
        m_tail = [UpscaleBlock(self._scale_factor, n_feats),
                  DefaultConv2d(in_channels=n_feats, out_channels=self._channels, kernel_size=kernel_size)]
        self.head = nn.Sequential(*m_head)
        self.body = nn.Sequential(*m_body)
        self.tail = nn.Sequential(*m_tail)

    def body_nhwc(self, f):
        """The residual trunk on NHWC features (edsr.py:44-47): `res = body(x); res += x`.  Its own method so that bench.py can time the
        trunk's launches in the order, with the buffers and in the autograd mode of a training step (`roofline.in_step`)."""
        blocks = list(self.body)[:-1]
        tail = self.body[-1]
        specs = [blk.plain_convs() for blk in blocks]
        if all(sp is not None for sp in specs) and len({blk.res_scale for blk in blocks}) == 1:
            # a batch that fills the chip with whole images: the whole trunk is ONE launch per direction (ops.ResTrunkFn)
            if ops.res_trunk_ok(f, specs, (tail.weight, tail.bias)):
                return ops.res_trunk(f, specs, (tail.weight, tail.bias), scale=blocks[0].res_scale)
        r = f
        for blk in blocks:
            r = ops.cut(blk.nhwc(r))
        return tail.nhwc(r, res=f)                               # body conv fused with `res += x` (edsr.py:46-47)

    def body_conv_launches(self):
        """(3x3 F -> F convolutions of the trunk per forward pass, F)"""
        return 2 * (len(self.body) - 1) + 1, self.body[-1].out_channels

    def forward(self, x):
        """NCHW float in [0,1] -> NCHW fp32, x scale_factor (edsr.py:40-54)."""
        with ops.forward_scope(self._pack_group()):
            rgb = self._channels == 3
            f = ops.head_conv(x, self.head[0].weight, self.head[0].bias, self.sub_mean.neg_shift() if rgb else None,
                              self.compute_dtype)
            f = ops.cut(f, keep=True)                            # (segment boundaries: identity unless ops.record_segments is active)
            r = self.body_nhwc(f)
            # upsampler (PixelShuffle = the conv's store addressing) + tail conv + add_mean; the last stage and the tail conv as ONE
            # collapsed 5x5 convolution on the 16-bit path (common.upscale_tail)
            up = list(self.tail[0])
            return upscale_tail(r, [(c, p.upscale_factor) for c, p in zip(up[0::2], up[1::2])], self.tail[1],
                                post_add=self.add_mean.shift() if rgb else None)
