"""Building blocks with the reference's names, constructor arguments and state_dict layout
(models/common.py:1-139), computing on the MI355X HIP path.

Inside a model, tensors are NHWC `[N,H,W,C]` in the model's compute dtype (see ops.py); these
modules therefore take/return NHWC tensors, NOT the reference's NCHW.  The NCHW contract lives
at `SRModel.forward()`, which converts at the head and tail convs.  Parameter creation order and
initialisation are exactly torch's `nn.Conv2d` (the classes derive from it), so
`torch.manual_seed(s); Model(...)` reproduces the reference's weights bit for bit.
"""
from math import log2

import torch
from torch import nn

from .. import ops


class DefaultConv2d(nn.Conv2d):
    """Conv2d that keeps H and W ('same' zero padding, stride 1).  Reference: common.py:7-30."""

    def __init__(self, kernel_size, padding='same', **kwargs):
        if isinstance(padding, str):
            lower_padding = padding.lower()
            assert lower_padding in ('valid', 'same')
            if lower_padding == 'valid':
                padding = 0
            elif isinstance(kernel_size, int):
                padding = kernel_size // 2
            else:
                padding = tuple(k // 2 for k in kernel_size)
        super().__init__(kernel_size=kernel_size, padding=padding, **kwargs)
        k = self.kernel_size[0]
        if self.kernel_size != (k, k) or self.padding != (k // 2, k // 2) or self.stride != (1, 1) \
                or self.dilation != (1, 1) or self.groups != 1:
            raise NotImplementedError("the HIP path implements square, stride-1, 'same' convolutions (all the hot path uses)")

    def forward(self, x, res=None, scale=1.0, ps_r=0):
        return ops.conv(x, self.weight, self.bias, res=res, scale=scale, ps_r=ps_r)


class MeanShift(nn.Conv2d):
    """Frozen 1x1 conv: W = I/std, b = sign*range*mean/std.  Reference: common.py:58-71.

    The parameters exist (and round-trip through state_dict) but the shift is folded into the
    neighbouring HIP kernel: `shift()` returns the per-channel bias after checking W == I."""

    def __init__(self, rgb_range=1, rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1.0, 1.0, 1.0), sign=-1):
        super().__init__(3, 3, kernel_size=1)
        std = torch.Tensor(rgb_std)
        self.weight.data = torch.eye(3).view(3, 3, 1, 1) / std.view(3, 1, 1, 1)
        self.bias.data = sign * rgb_range * torch.Tensor(rgb_mean) / std
        for p in self.parameters():
            p.requires_grad = False
        self._checked = None

    def shift(self):
        key = (self.weight._version, self.weight.data_ptr())
        if self._checked != key:
            if not torch.equal(self.weight.detach().cpu().view(3, 3), torch.eye(3)):
                raise NotImplementedError("MeanShift with rgb_std != 1 is not on the fused path")
            self._checked = key
        return self.bias

    def forward(self, x):
        raise RuntimeError("MeanShift is fused into the head/tail HIP kernels; call shift()")


class ResBlock(nn.Module):
    """conv -> ReLU -> conv, * res_scale, += x.  Reference: common.py:74-109 (n_conv_layers convs,
    activation between them, no norm on the hot path)."""

    def __init__(self, conv=DefaultConv2d, n_feats=64, kernel_size=3, n_conv_layers=2, bias=True, norm=None,
                 act=nn.ReLU(True), res_scale=1.):
        super().__init__()
        if norm is not None or not (act is None or isinstance(act, nn.ReLU)):
            raise NotImplementedError("the HIP path implements ReLU / no norm (what EDSR uses)")
        m = []
        for i in range(n_conv_layers):
            m.append(conv(in_channels=n_feats, out_channels=n_feats, kernel_size=kernel_size, bias=bias))
            if act is not None and i < n_conv_layers - 1:
                m.append(act)
        self.body = nn.Sequential(*m)
        self.res_scale = res_scale
        self._has_act = act is not None

    def forward(self, x):
        convs = [m for m in self.body if isinstance(m, nn.Conv2d)]
        relus = [self._has_act and i < len(convs) - 1 for i in range(len(convs))]
        return ops.conv_chain(x, [(c.weight, c.bias) for c in convs], relus, scale=self.res_scale)


class UpscaleBlock(nn.Sequential):
    """[conv3x3(F -> F r^2), PixelShuffle(r)] x int(log2(s)).  Reference: common.py:112-139.
    The PixelShuffle modules are kept (index/layout compatibility) but the shuffle is the conv's store."""

    def __init__(self, scale_factor=4, n_feats=64, kernel_size=3, act=None):
        assert scale_factor in {2, 3, 4, 8}
        if act is not None:
            raise NotImplementedError("UpscaleBlock activation is not used on the hot path")
        layers = []
        for _ in range(int(log2(scale_factor))):
            r = 2 if scale_factor % 2 == 0 else 3
            layers += [DefaultConv2d(in_channels=n_feats, out_channels=n_feats * r * r, kernel_size=kernel_size),
                       nn.PixelShuffle(r)]
        super().__init__(*layers)

    def forward(self, x):
        mods = list(self)
        for conv, ps in zip(mods[0::2], mods[1::2]):
            x = conv(x, ps_r=ps.upscale_factor)
        return x
