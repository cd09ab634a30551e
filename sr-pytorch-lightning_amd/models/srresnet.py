"""SRResNet on the HIP path.  Reference: models/srresnet.py:9-36 (same ctor, same state_dict keys -- including the
duplicate keys of the BatchNorm / PReLU instances that the reference's `ResBlock` registers twice, common.py:94-100)."""
from typing import Any

import torch.nn as nn

from .. import ops
from .common import BasicBlock, DefaultConv2d, ResBlock, UpscaleBlock
from .srmodel import SRModel


class SRResNet(SRModel):
    def __init__(self, n_resblocks: int = 16, n_feats: int = 64, **kwargs: dict[str, Any]):
        super().__init__(**kwargs)
        self.head = BasicBlock(in_channels=self._channels, out_channels=n_feats, kernel_size=9, act=nn.PReLU())
        m_body = [ResBlock(n_feats=n_feats, kernel_size=3, n_conv_layers=2, norm=nn.BatchNorm2d(n_feats), act=nn.PReLU())
                  for _ in range(n_resblocks)]
        m_body.append(BasicBlock(in_channels=n_feats, out_channels=n_feats, kernel_size=3, norm=nn.BatchNorm2d(n_feats), act=None))
        self.body = nn.Sequential(*m_body)
        m_tail = [UpscaleBlock(self._scale_factor, n_feats=n_feats, act=nn.PReLU()),
                  DefaultConv2d(in_channels=n_feats, out_channels=self._channels, kernel_size=9)]
        self.tail = nn.Sequential(*m_tail)

    def forward(self, x):
        """srresnet.py:32-36: head (9x9 conv + PReLU), body + skip, upsampler, 9x9 conv.  NCHW float in, NCHW fp32 out."""
        with ops.forward_scope(self._pack_group()):
            hc, ha = self.head[0], self.head[1]
            f = ops.prelu(ops.head_conv(x, hc.weight, hc.bias, None, self.compute_dtype), ha.weight)   # 9x9 over 3 channels: boundary im2col + 1x1 MFMA
            r = f
            blocks = list(self.body)
            for blk in blocks[:-1]:
                r = blk.nhwc(r)
            r = blocks[-1].nhwc(r, res=f)                       # conv + BatchNorm, `+ x` fused into the BatchNorm apply
            r = self.tail[0].nhwc(r)
            t = self.tail[1]
            y = ops.conv_general(r, t.weight, t.bias, stride=1, pad=t.kernel_size[0] // 2)               # 9x9 over 64 channels: the direct large-kernel kernels in 16-bit (csrc/conv_lk.hip), im2col + 1x1 MFMA in fp32
            return ops.nhwc_to_nchw(y, self._channels)
