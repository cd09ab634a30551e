"""WDSR on the HIP path.  Reference: models/wdsr.py:9-117 (same ctor, same state_dict keys)."""
from typing import Any

import torch
import torch.nn as nn

from .. import ops
from .common import _NCHWContract
from .srmodel import SRModel


def _wn_weight(m):
    """Legacy nn.utils.weight_norm re-parametrisation w = g * v / ||v|| (wdsr.py:62), kept in PyTorch
    (parameter-sized work); the resulting OIHW weight feeds the HIP conv and receives its gradient."""
    return torch._weight_norm(m.weight_v, m.weight_g, 0)


class _Block_A(_NCHWContract, nn.Module):
    """3x3 (F -> 4F) -> ReLU -> 3x3 (4F -> F), * res_scale, += x (wdsr.py:9-27)."""

    def __init__(self, n_feats, kernel_size, wn, act=nn.ReLU(True), res_scale=1):
        super().__init__()
        self.res_scale = res_scale
        block_feats = 4 * n_feats
        self.body = nn.Sequential(wn(nn.Conv2d(n_feats, block_feats, kernel_size, padding=kernel_size // 2)), act,
                                  wn(nn.Conv2d(block_feats, n_feats, kernel_size, padding=kernel_size // 2)))

    def _cout(self):
        return self.body[0].in_channels

    def wn_convs(self):
        return [self.body[0], self.body[2]]

    def nhwc(self, x, ws=None):
        c1, c2 = self.body[0], self.body[2]
        w1, w2 = ws if ws is not None else (_wn_weight(c1), _wn_weight(c2))
        return ops.conv_chain(x, [(w1, c1.bias), (w2, c2.bias)], [True, False], scale=self.res_scale)


class _Block_B(_NCHWContract, nn.Module):
    """1x1 (F -> 6F) -> ReLU -> 1x1 (6F -> int(.8F)) -> 3x3 (-> F), * res_scale, += x (wdsr.py:30-51)."""

    def __init__(self, n_feats, kernel_size, wn, act=nn.ReLU(True), res_scale=1):
        super().__init__()
        self.res_scale = res_scale
        expand, linear = 6, 0.8
        self.body = nn.Sequential(wn(nn.Conv2d(n_feats, n_feats * expand, 1, padding=1 // 2)), act,
                                  wn(nn.Conv2d(n_feats * expand, int(n_feats * linear), 1, padding=1 // 2)),
                                  wn(nn.Conv2d(int(n_feats * linear), n_feats, kernel_size, padding=kernel_size // 2)))

    def _cout(self):
        return self.body[0].in_channels

    def wn_convs(self):
        return [self.body[0], self.body[2], self.body[3]]

    def nhwc(self, x, ws=None):
        c1, c2, c3 = self.body[0], self.body[2], self.body[3]
        w1, w2, w3 = ws if ws is not None else (_wn_weight(c1), _wn_weight(c2), _wn_weight(c3))
        return ops.wdsr_block_b(x, [(w1, c1.bias), (w2, c2.bias), (w3, c3.bias)], scale=self.res_scale)


class WDSR(SRModel):
    # all 51 effective weights come from ONE autograd node (ops.WeightNormGroup): a backward pass cut into segments (trainer.
    # OverlappedGraphStep) would run that node in the first segment's pass with the top blocks' gradients only and find its saved
    # tensors freed in the second -- the multi-rank step keeps WDSR's backward in one piece (trainer.auto_segments)
    supports_backward_segments = False

    def __init__(self, type: str = 'B', n_feats: int = 128, n_resblocks: int = 16, res_scale: int = 1, **kwargs: dict[str, Any]):
        super().__init__(**kwargs)
        kernel_size = 3

        def wn(x):
            return nn.utils.weight_norm(x)

        if self._channels == 3:
            # plain attribute, not a buffer, as in the reference (wdsr.py:66-67)
            self.rgb_mean = torch.FloatTensor([0.4488, 0.4371, 0.4040]).view([1, 3, 1, 1])
        head = [wn(nn.Conv2d(self._channels, n_feats, 3, padding=3 // 2))]
        block = _Block_A if type == 'A' else _Block_B
        body = [block(n_feats, kernel_size, act=nn.ReLU(True), res_scale=res_scale, wn=wn) for _ in range(n_resblocks)]
        out_feats = self._scale_factor * self._scale_factor * self._channels
        tail = [wn(nn.Conv2d(n_feats, out_feats, 3, padding=3 // 2)), nn.PixelShuffle(self._scale_factor)]
        skip = [wn(nn.Conv2d(3, out_feats, 5, padding=5 // 2)), nn.PixelShuffle(self._scale_factor)]
        self.head = nn.Sequential(*head)
        self.body = nn.Sequential(*body)
        self.tail = nn.Sequential(*tail)
        self.skip = nn.Sequential(*skip)

    def forward(self, x):
        """wdsr.py:102-117: x - mean; s = PS(skip(x)); x = PS(tail(body(head(x)))); x += s; x + mean."""
        # all 51 effective weights w = g * v / ||v|| in ONE launch, into a buffer with a stable address (ops.WeightNormGroup):
        # they join the grouped pack launch and the grouped weight gradients like leaf parameters do
        grp = self.__dict__.get("_srk_wn")
        if grp is None:
            convs = [self.skip[0], self.head[0]]
            for blk in self.body:
                convs += blk.wn_convs()
            grp = self.__dict__["_srk_wn"] = ops.WeightNormGroup(convs + [self.tail[0]])
        ws = grp.weights()
        with ops.forward_scope(self._pack_group()):
            mean = None
            if self._channels == 3:
                self.rgb_mean = self.rgb_mean.to(x.device)
                mean = self.rgb_mean.view(3).contiguous()
            r = self._scale_factor
            s = ops.skip_conv(x, ws[0], self.skip[0].bias, mean, r, self.compute_dtype)
            f = ops.head_conv(x, ws[1], self.head[0].bias, mean, self.compute_dtype)
            s, f = ops.cut(s, f, keep=True)
            k = 2
            for blk in self.body:
                nb = len(blk.wn_convs())
                f = ops.cut(blk.nhwc(f, ws[k:k + nb]))
                k += nb
            return ops.tail_conv(f, ws[k], self.tail[0].bias, res=s, post_add=mean, ps_r=r)
