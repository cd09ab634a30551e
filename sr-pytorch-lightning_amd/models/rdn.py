"""RDN on the HIP path.  Reference: models/rdn.py:9-111 (same ctor, same state_dict keys)."""
from typing import Any

import torch
import torch.nn as nn

from .. import ops
from .common import _NCHWContract, upscale_tail
from .srmodel import SRModel


class _RDB_Conv(nn.Module):
    """conv3x3(Cin -> G) + ReLU appended to the input (rdn.py:9-21).  Parameters only: the dense block's
    fused Function writes the G channels into a slice of the block's buffer instead of torch.cat."""

    def __init__(self, inChannels, growRate, kSize=3):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(inChannels, growRate, kSize, padding=(kSize - 1) // 2, stride=1), nn.ReLU())


class _RDB(_NCHWContract, nn.Module):
    """rdn.py:24-40"""

    def __init__(self, growRate0, growRate, nConvLayers, kSize=3):
        super().__init__()
        G0, G, C = growRate0, growRate, nConvLayers
        self.convs = nn.Sequential(*[_RDB_Conv(G0 + c * G, G) for c in range(C)])
        self.LFF = nn.Conv2d(G0 + C * G, G0, 1, padding=0, stride=1)

    def _cout(self):
        return self.LFF.out_channels

    def nhwc(self, x, dest=None):
        return ops.rdb(x, [(m.conv[0].weight, m.conv[0].bias) for m in self.convs], (self.LFF.weight, self.LFF.bias), dest=dest)


class RDN(SRModel):
    def __init__(self, rdn_config: str = 'B', G0: int = 64, kernel_size: int = 3, **kwargs: dict[str, Any]):
        super().__init__(**kwargs)
        self.D, C, G = {'A': (20, 6, 32), 'B': (16, 8, 64)}[rdn_config]
        pad = (kernel_size - 1) // 2
        self.SFENet1 = nn.Conv2d(self._channels, G0, kernel_size, padding=pad, stride=1)
        self.SFENet2 = nn.Conv2d(G0, G0, kernel_size, padding=pad, stride=1)
        self._RDBs = nn.ModuleList()
        for _ in range(self.D):
            self._RDBs.append(_RDB(growRate0=G0, growRate=G, nConvLayers=C))
        self.GFF = nn.Sequential(nn.Conv2d(self.D * G0, G0, 1, padding=0, stride=1),
                                 nn.Conv2d(G0, G0, kernel_size, padding=pad, stride=1))
        s = self._scale_factor
        if s == 2 or s == 3:
            self.UPNet = nn.Sequential(nn.Conv2d(G0, G * s * s, kernel_size, padding=pad, stride=1),
                                       nn.PixelShuffle(s),
                                       nn.Conv2d(G, 3, kernel_size, padding=pad, stride=1))
        elif s == 4:
            self.UPNet = nn.Sequential(nn.Conv2d(G0, G * 4, kernel_size, padding=pad, stride=1),
                                       nn.PixelShuffle(2),
                                       nn.Conv2d(G, G * 4, kernel_size, padding=pad, stride=1),
                                       nn.PixelShuffle(2),
                                       nn.Conv2d(G, self._channels, kernel_size, padding=pad, stride=1))
        else:
            raise ValueError("scale must be 2 or 3 or 4.")

    def forward(self, x):
        """rdn.py:99-111.  No MeanShift in this model."""
        with ops.forward_scope(self._pack_group()):
            f1 = ops.cut(ops.head_conv(x, self.SFENet1.weight, self.SFENet1.bias, None, self.compute_dtype), keep=True)
            x = ops.conv(f1, self.SFENet2.weight, self.SFENet2.bias)
            # `torch.cat(RDBs_out, 1)` (rdn.py:108) without the copy: every block writes its output into its channel slice of ONE
            # buffer, which the 1x1 global feature fusion then reads whole
            outs, cat = [], ops.SliceBuffer(len(self._RDBs))
            for i, blk in enumerate(self._RDBs):
                x = ops.cut(blk.nhwc(x, dest=(cat, i)), keep=True)     # every block output also feeds the global fusion
                outs.append(x)
            x = ops.conv(ops.concat_slices(cat, outs), self.GFF[0].weight, self.GFF[0].bias)    # 1x1 over D*G0 channels
            x = ops.conv(x, self.GFF[1].weight, self.GFF[1].bias, res=f1)                  # `x += f__1`
            mods = list(self.UPNet)
            return upscale_tail(x, [(c, p.upscale_factor) for c, p in zip(mods[0:-1:2], mods[1:-1:2])], mods[-1])
