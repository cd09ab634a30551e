"""SRCNN (models/srcnn.py:9-27): BASELINE config 0, the reference's own CPU-runnable plumbing case.

NOT part of the HIP hot path: bicubic interpolation + 9x9/1x1/5x5 convs in
plain PyTorch on whatever device the tensors live on.  It exists so that the `SRModel` surface
(ctor, training_step, configure_optimizers) can be exercised without a GPU.
"""
from typing import Any

import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from .srmodel import SRModel


class SRCNN(SRModel):
    def __init__(self, **kwargs: dict[str, Any]):
        super().__init__(**kwargs)
        self._net = nn.Sequential(nn.Conv2d(self._channels, 64, 9, padding=4), nn.ReLU(True),
                                  nn.Conv2d(64, 32, 1, padding=0), nn.ReLU(True),
                                  nn.Conv2d(32, self._channels, 5, padding=2))

    def forward(self, x):
        x = F.interpolate(x, scale_factor=self._scale_factor, mode='bicubic')
        for m in self._net:
            x = m(x)
            if isinstance(m, nn.ReLU):
                x = ops.cut(x)              # segment boundary (identity unless ops.record_segments is active)
        return x
