"""Building blocks with the reference's names, constructor arguments and state_dict layout
(models/common.py:1-139), computing on the MI355X HIP path.

Inside a model, tensors are NHWC `[N,H,W,C]` in the model's compute dtype (see ops.py): the models call the
blocks' `.nhwc(...)` methods.  `forward(x)` keeps the REFERENCE's contract -- NCHW float in, NCHW float out -- so a user
subclass of `SRModel` that composes these blocks on NCHW tensors (README.md:97-101 of the reference) works unchanged:
it converts at the block boundary (two layout kernels) and computes in `self.compute_dtype` (fp32 unless the attribute
is set on the block).  Parameter creation order and
initialisation are exactly torch's `nn.Conv2d` (the classes derive from it), so
`torch.manual_seed(s); Model(...)` reproduces the reference's weights bit for bit.
"""
from math import log2

import torch
from torch import nn

from .. import ops


class _NCHWContract:
    """forward(x NCHW float) -> NCHW fp32 around the block's NHWC implementation (`nhwc`); `_cout()` = real output channels."""

    def forward(self, x, *args, **kwargs):
        dt = getattr(self, "compute_dtype", torch.float32)
        y = self.nhwc(ops.nchw_to_nhwc(x, dt), *args, **kwargs)
        return ops.nhwc_to_nchw(y, self._cout())


class DefaultConv2d(_NCHWContract, nn.Conv2d):
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

    def _cout(self):
        return self.out_channels

    def nhwc(self, x, res=None, scale=1.0, ps_r=0, link=None):
        return ops.conv(x, self.weight, self.bias, res=res, scale=scale, ps_r=ps_r, link=link)


class BasicBlock(_NCHWContract, nn.Sequential):
    """Conv2d [+ norm] [+ activation].  Reference: common.py:33-56 (SRResNet's head and body tail, srresnet.py:13-21)."""

    def __init__(self, in_channels=64, out_channels=64, kernel_size=3, bias=True, conv=DefaultConv2d, norm=None, act=nn.ReLU(True)):
        m = [conv(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, bias=bias)]
        if norm is not None:
            m.append(norm)
        if act is not None:
            m.append(act)
        super().__init__(*m)

    def _cout(self):
        return self[0].out_channels

    def nhwc(self, x, res=None):
        """x: NHWC features.  `res` is added after the norm (fused into the BatchNorm apply kernel when there is one)."""
        mods = list(self)
        conv = mods[0]
        norm = next((m for m in mods[1:] if isinstance(m, nn.BatchNorm2d)), None)
        act = next((m for m in mods[1:] if isinstance(m, (nn.PReLU, nn.ReLU))), None)
        if conv.kernel_size[0] in (1, 3):
            r = conv.nhwc(x, res=res if (norm is None and act is None) else None)
        else:
            r = ops.conv_general(x, conv.weight, conv.bias, stride=1, pad=conv.kernel_size[0] // 2)
        done_res = norm is None and act is None and conv.kernel_size[0] in (1, 3)
        if norm is not None and isinstance(act, nn.PReLU):
            r = ops.batch_norm_prelu(r, norm, act.weight)         # one unit: the BatchNorm's output is never stored
        else:
            if norm is not None:
                r = ops.batch_norm(r, norm, res=res if act is None else None)
                done_res = done_res or act is None
            if isinstance(act, nn.PReLU):
                r = ops.prelu(r, act.weight)
            elif act is not None:
                r = torch.relu(r)
        if res is not None and not done_res:
            r = r + res
        return r


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

    def neg_shift(self):
        """-shift(), cached (the parameter is frozen; a load_state_dict or an in-place edit bumps its version): the head kernels
        SUBTRACT this vector, and a `-bias` per step is one more parameter-sized launch."""
        b = self.shift()
        key = (b._version, b.data_ptr())
        if self.__dict__.get("_neg_key") != key:
            self.__dict__["_neg"], self.__dict__["_neg_key"] = (-b.detach()), key
        return self.__dict__["_neg"]

    def forward(self, x):
        """The reference's op on an NCHW image (3 multiply-adds per pixel; inside the models the shift is fused into the
        head / tail kernels instead)."""
        return x + self.shift().view(1, 3, 1, 1).to(x.dtype)


class ResBlock(_NCHWContract, nn.Module):
    """conv [-> norm] -> act -> conv [-> norm], * res_scale, += x.  Reference: common.py:74-109.  The SAME `norm` / `act`
    module instance is appended after every conv, exactly like the reference (shared BatchNorm parameters and running
    statistics inside a block: srresnet.py:16-19 passes one nn.BatchNorm2d and one nn.PReLU per block)."""

    def __init__(self, conv=DefaultConv2d, n_feats=64, kernel_size=3, n_conv_layers=2, bias=True, norm=None,
                 act=nn.ReLU(True), res_scale=1.):
        super().__init__()
        if not (norm is None or isinstance(norm, nn.BatchNorm2d)) or not (act is None or isinstance(act, (nn.ReLU, nn.PReLU))):
            raise NotImplementedError("the HIP path implements BatchNorm2d / no norm and ReLU / PReLU / no activation")
        m = []
        for i in range(n_conv_layers):
            m.append(conv(in_channels=n_feats, out_channels=n_feats, kernel_size=kernel_size, bias=bias))
            if norm is not None:
                m.append(norm)
            if act is not None and i < n_conv_layers - 1:
                m.append(act)
        self.body = nn.Sequential(*m)
        self.res_scale = res_scale
        self._has_act = act is not None

    def _cout(self):
        return self.body[0].out_channels

    def plain_convs(self):
        """[(w1, b1), (w2, b2)] when the block is conv -> ReLU -> conv (EDSR's form), else None: what ops.res_trunk takes."""
        mods = list(self.body)
        if len(mods) == 3 and isinstance(mods[0], nn.Conv2d) and isinstance(mods[1], nn.ReLU) and isinstance(mods[2], nn.Conv2d):
            return [(mods[0].weight, mods[0].bias), (mods[2].weight, mods[2].bias)]
        return None

    def nhwc(self, x):
        mods = list(self.body)
        convs = [m for m in mods if isinstance(m, nn.Conv2d)]
        plain = all(isinstance(m, (nn.Conv2d, nn.ReLU)) for m in mods)
        if plain:              # EDSR: the fused conv chain (ReLU, * res_scale and += x are conv epilogues)
            relus = [self._has_act and i < len(convs) - 1 for i in range(len(convs))]
            return ops.conv_chain(x, [(c.weight, c.bias) for c in convs], relus, scale=self.res_scale)
        # SRResNet: conv -> BatchNorm -> PReLU -> conv -> BatchNorm (+ x fused into the last BatchNorm apply)
        r = x
        fused_res = False
        skip = False
        # the residual add rides in the last BatchNorm's apply launch; its gradient rides in the first conv's data-gradient launch
        link = ops.ResLink() if (ops._RES_LINK and isinstance(mods[-1], nn.BatchNorm2d) and self.res_scale == 1 and isinstance(mods[0], nn.Conv2d)
                                 and mods[0].kernel_size[0] == 3) else None
        for i, m in enumerate(mods):
            last = i == len(mods) - 1
            if skip:                                   # (the PReLU that rode in the BatchNorm before it)
                skip = False
                continue
            if isinstance(m, nn.Conv2d):
                r = m.nhwc(r, link=link if i == 0 else None)
            elif isinstance(m, nn.BatchNorm2d) and not last and isinstance(mods[i + 1], nn.PReLU):
                r = ops.batch_norm_prelu(r, m, mods[i + 1].weight)
                skip = True
            elif isinstance(m, nn.BatchNorm2d):
                if last and self.res_scale == 1:
                    r = ops.batch_norm(r, m, res=x, link=link)
                    fused_res = True
                else:
                    r = ops.batch_norm(r, m)
            elif isinstance(m, nn.PReLU):
                r = ops.prelu(r, m.weight)
            else:
                r = torch.relu(r)
        if not fused_res:
            r = r * self.res_scale + x
        return r


class UpscaleBlock(_NCHWContract, nn.Sequential):
    """[conv3x3(F -> F r^2), PixelShuffle(r) [, act]] x int(log2(s)).  Reference: common.py:112-139.
    The PixelShuffle modules are kept (index/layout compatibility) but the shuffle is the conv's store; `act` (SRResNet:
    one shared nn.PReLU, srresnet.py:26-27) runs as its own HIP kernel behind it."""

    def __init__(self, scale_factor=4, n_feats=64, kernel_size=3, act=None):
        assert scale_factor in {2, 3, 4, 8}
        if not (act is None or isinstance(act, (nn.PReLU, nn.ReLU))):
            raise NotImplementedError("UpscaleBlock activation: PReLU / ReLU / none")
        layers = []
        for _ in range(int(log2(scale_factor))):
            r = 2 if scale_factor % 2 == 0 else 3
            layers += [DefaultConv2d(in_channels=n_feats, out_channels=n_feats * r * r, kernel_size=kernel_size),
                       nn.PixelShuffle(r)]
            if act is not None:
                layers.append(act)
        super().__init__(*layers)

    def _cout(self):
        return self[0].in_channels

    def nhwc(self, x, stop_before_last=False):
        """stop_before_last: run every stage but the last and return (x, last conv, its PixelShuffle factor) -- the caller
        (`upscale_tail`) may run that stage and the tail conv as one collapsed 5x5 convolution.  Only without an activation."""
        mods = list(self)
        i = 0
        while i < len(mods):
            conv, ps = mods[i], mods[i + 1]
            if stop_before_last and i + 2 >= len(mods):
                return x, conv, ps.upscale_factor
            x = conv.nhwc(x, ps_r=ps.upscale_factor)
            i += 2
            if i < len(mods) and isinstance(mods[i], (nn.PReLU, nn.ReLU)):
                x = ops.prelu(x, mods[i].weight) if isinstance(mods[i], nn.PReLU) else torch.relu(x)
                i += 1
        return x


def upscale_tail(x, stages, tail, post_add=None):
    """The upsampler stages `[(conv, ps_r), ...]` followed by the tail conv -> NCHW fp32 image (edsr.py:48-52, rcan.py:102-104,
    rdn.py:85-95 / 110).  Nothing non-linear sits between the LAST stage and the tail conv, so on the 16-bit path the two run as one
    5x5 convolution whose weights are built from theirs (`ops.hr_tail`: no C-channel tensor at the output resolution, 8x fewer
    multiply-adds, same function and gradients up to rounding); everything else -- fp32, other shapes, PixelShuffle(3) -- keeps the
    layer-by-layer form."""
    def stage(x, conv, r):
        return conv.nhwc(x, ps_r=r) if hasattr(conv, "nhwc") else ops.conv(x, conv.weight, conv.bias, ps_r=r)
    for conv, r in stages[:-1]:
        x = stage(x, conv, r)
    if stages:
        conv, r = stages[-1]
        if conv.kernel_size == (3, 3) and tail.kernel_size == (3, 3) and ops.hr_tail_ok(x, conv.weight, tail.weight, r):
            return ops.hr_tail(x, conv.weight, conv.bias, tail.weight, tail.bias, post_add=post_add)
        x = stage(x, conv, r)
    return ops.tail_conv(x, tail.weight, tail.bias, post_add=post_add)
