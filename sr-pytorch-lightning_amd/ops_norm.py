"""BatchNorm2d / PReLU and the per-channel statistics / apply kernels behind them (SRResNet, D-DBPN; csrc/generic.hip).
Part of `ops` (re-exported there): split out of ops.py in round 6 by op family."""
import os

import torch

from . import _lib as L
from .ops import _DT, _f32c, _grad_slot, _knob, _need_gpu, _pass_slot, _pitch, _ptr, _remember_pass_grad, _stream      # (ops.py imports this module at its END: these exist by then)


def chan_sums(x, y=None, mode=None, shift=None):
    """Per-channel sums over all pixels of NHWC `x`: (sum x, sum x^2); with y: mode 1 (sum y, sum x*y) or 2 (sum over x <= 0
    of x*y, 0).  `shift` [C] fp32: x is centred (x - shift[c]) first.  srk_chan_stats partials added in block order
    (fp32, reproducible)."""
    _need_gpu(x)
    c = x.shape[-1]
    P = x.numel() // c
    mode = (0 if y is None else 1) if mode is None else mode
    if P == 0:
        z = torch.zeros(c, dtype=torch.float32, device=x.device)
        return z, z.clone()
    nb = L.load().srk_chan_stats_blocks(P)
    part = torch.empty((nb, 2, c), dtype=torch.float32, device=x.device)
    L.call("srk_chan_stats", L.ChanStatsArgs(x=x.data_ptr(), x_pitch=_pitch4(x), x_coff=0, y=_ptr(y), y_pitch=0 if y is None else _pitch4(y), y_coff=0,
                                             P=P, C=c, mode=mode, partial=part.data_ptr(), dtype=_DT[x.dtype], shift=_ptr(shift), shift_out=0), _stream())
    tot = part.sum(0)
    return tot[0], tot[1]


def _pitch4(t):
    return _pitch(t) if t.dim() == 4 else t.shape[-1]


def _gate_fields(gate):
    """ChanStatsArgs fields of the optional PReLU input-gradient output: gate = (out tensor, fp32 slope [1] or [C]); or, for mode 3
    (BatchNorm backward behind a PReLU), gate = ("bn", gate_a, gate_d, slope, partial2)."""
    if gate is None:
        return dict(gate_out=0, gate_pitch=0, slope=0, slope_stride=0)
    if gate[0] == "bn":
        _, ga, gd, sl, p2 = gate
        return dict(gate_out=0, gate_pitch=0, slope=sl.data_ptr(), slope_stride=0 if sl.numel() == 1 else 1,
                    gate_a=ga.data_ptr(), gate_d=gd.data_ptr(), partial2=p2.data_ptr())
    out, sl = gate
    return dict(gate_out=out.data_ptr(), gate_pitch=_pitch4(out), slope=sl.data_ptr(), slope_stride=0 if sl.numel() == 1 else 1)


def chan_partials(x, y=None, mode=0, shift=None, shift_out=None, gate=None):
    """The per-block partial sums of srk_chan_stats [blocks][2][C] (see chan_sums), left unsummed for srk_chan_finalize."""
    _need_gpu(x)
    c = x.shape[-1]
    P = x.numel() // c
    nb = L.load().srk_chan_stats_blocks(P)
    part = torch.empty((nb, 2, c), dtype=torch.float32, device=x.device)
    L.call("srk_chan_stats", L.ChanStatsArgs(x=x.data_ptr(), x_pitch=_pitch4(x), x_coff=0, y=_ptr(y), y_pitch=0 if y is None else _pitch4(y), y_coff=0,
                                             P=P, C=c, mode=mode, partial=part.data_ptr(), dtype=_DT[x.dtype], shift=_ptr(shift), shift_out=_ptr(shift_out),
                                             **_gate_fields(gate)), _stream())
    return part


_COUNTERS = {}
_FUSE_MAX_BLOCKS = int(_knob("SRK_CHAN_FUSE_MAX_BLOCKS", "160"))


def _arrival_counter(device):
    """One int32 of device memory per call, handed out round-robin from a zeroed pool: srk_chan_stats_finalize's arrival counter (the
    kernel leaves it 0).  Launches that could overlap (other streams, parallel branches of a replayed graph) are many calls apart."""
    st = _COUNTERS.get(device)
    if st is None:
        st = _COUNTERS[device] = [torch.zeros(4096, dtype=torch.int32, device=device), 0]
    st[1] = (st[1] + 1) % 4096
    return st[0].data_ptr() + 4 * st[1]


def chan_reduce(x, y, smode, shift, fmode, rows, *, M=1.0, creal=None, eps=0.0, momentum=0.0, mean=None, invstd=None, gamma=None, weight=None,
                bias=None, running_mean=None, running_var=None, total=False, nbt=None, dgamma_acc=None, dbeta_acc=None, shift_out=None, gate=None,
                bn_gate=None, total2=False, dslope_acc=None):
    """chan_finalize(chan_partials(x, y, smode, shift), fmode, rows, ...) as ONE launch (srk_chan_stats_finalize: the block that
    finishes last does the [C]-sized step)."""
    _need_gpu(x)
    c = x.shape[-1]
    P = x.numel() // c
    nb = L.load().srk_chan_stats_blocks(P)
    p2 = None
    if bn_gate is not None:              # smode 3: (gate_a, gate_d, slope) -> third partial sums (the PReLU slope gradient)
        p2 = torch.empty((nb, c), dtype=torch.float32, device=x.device)
        gate = ("bn", bn_gate[0], bn_gate[1], bn_gate[2], p2)
    if nb > _FUSE_MAX_BLOCKS:            # many blocks: their arrival counts (same-address atomics) would take longer than the launch they save
        return chan_finalize(chan_partials(x, y, smode, shift, shift_out, gate), fmode, rows, partial2=p2, total2=total2, dslope_acc=dslope_acc, M=M, creal=creal, eps=eps, momentum=momentum, mean=mean, invstd=invstd,
                             gamma=gamma, weight=weight, bias=bias, running_mean=running_mean, running_var=running_var, total=total,
                             nbt=nbt, dgamma_acc=dgamma_acc, dbeta_acc=dbeta_acc)
    part = torch.empty((nb, 2, c), dtype=torch.float32, device=x.device)
    out = torch.empty((rows, c), dtype=torch.float32, device=x.device)
    sa = L.ChanStatsArgs(x=x.data_ptr(), x_pitch=_pitch4(x), x_coff=0, y=_ptr(y), y_pitch=0 if y is None else _pitch4(y), y_coff=0,
                         P=P, C=c, mode=smode, partial=part.data_ptr(), dtype=_DT[x.dtype], shift=_ptr(shift), shift_out=_ptr(shift_out),
                         **_gate_fields(gate))
    fa = L.ChanFinalizeArgs(
        partial=part.data_ptr(), nblocks=nb, C=c, Creal=c if creal is None else creal, mode=fmode, total=int(total),
        M=float(M), eps=float(eps), momentum=float(momentum), mean=_ptr(mean), invstd=_ptr(invstd), gamma=_ptr(gamma),
        weight=_ptr(weight), bias=_ptr(bias), running_mean=_ptr(running_mean), running_var=_ptr(running_var), out=out.data_ptr(),
        nbt=_ptr(nbt), dgamma_acc=_ptr(dgamma_acc), dbeta_acc=_ptr(dbeta_acc), partial2=_ptr(p2), total2=int(total2), dslope_acc=_ptr(dslope_acc))
    import ctypes as C
    L.check(L.load().srk_chan_stats_finalize(C.byref(sa), C.byref(fa), C.c_void_p(_arrival_counter(x.device)), C.c_void_p(_stream())),
            "srk_chan_stats_finalize")
    return out


def chan_finalize(part, mode, rows, *, M=1.0, creal=None, eps=0.0, momentum=0.0, mean=None, invstd=None, gamma=None, weight=None, bias=None,
                  running_mean=None, running_var=None, total=False, nbt=None, dgamma_acc=None, dbeta_acc=None, partial2=None, total2=False,
                  dslope_acc=None):
    """srk_chan_finalize on the partials of chan_partials: `rows` x [C] fp32 results (include/srk.h lists them per mode)."""
    nb, _, c = part.shape
    out = torch.empty((rows, c), dtype=torch.float32, device=part.device)
    L.call("srk_chan_finalize", L.ChanFinalizeArgs(
        partial=part.data_ptr(), nblocks=nb, C=c, Creal=c if creal is None else creal, mode=mode, total=int(total),
        M=float(M), eps=float(eps), momentum=float(momentum), mean=_ptr(mean), invstd=_ptr(invstd), gamma=_ptr(gamma),
        weight=_ptr(weight), bias=_ptr(bias), running_mean=_ptr(running_mean), running_var=_ptr(running_var), out=out.data_ptr(),
        nbt=_ptr(nbt), dgamma_acc=_ptr(dgamma_acc), dbeta_acc=_ptr(dbeta_acc), partial2=_ptr(partial2), total2=int(total2),
        dslope_acc=_ptr(dslope_acc)), _stream())
    return out


def chan_apply(x, *, y=None, z=None, a=None, b=None, d=None, slope=None, post_prelu=False, out=None, gate_a=None, gate_d=None):
    """out = post((a*x + b*y + d) * gate(z)) per channel (srk_chan_apply).  a/b/d/slope: fp32 [C] (slope may have 1 element).
    out: a tensor of x's shape to write (a channel-slice view of a wider NHWC buffer is fine), else a fresh one."""
    _need_gpu(x)
    c = x.shape[-1]
    P = x.numel() // c
    if out is None:
        out = torch.empty(x.shape, dtype=x.dtype, device=x.device)
    else:
        assert tuple(out.shape) == tuple(x.shape) and out.dtype == x.dtype
    if P == 0:
        return out
    sl = None if slope is None else _f32c(slope)

    def vec(v):
        if v is None:
            return None
        v = _f32c(v)
        return v if v.numel() == c else torch.nn.functional.pad(v, (0, c - v.numel()))
    a, b, d = vec(a), vec(b), vec(d)
    if sl is not None and sl.numel() not in (1, c):
        sl = torch.nn.functional.pad(sl, (0, c - sl.numel()))
    L.call("srk_chan_apply", L.ChanApplyArgs(
        x=x.data_ptr(), x_pitch=_pitch4(x), x_coff=0, y=_ptr(y), y_pitch=0 if y is None else _pitch4(y), y_coff=0,
        z=_ptr(z), z_pitch=0 if z is None else _pitch4(z), z_coff=0, a=_ptr(a), b=_ptr(b), d=_ptr(d),
        slope=_ptr(sl), slope_stride=0 if (sl is None or sl.numel() == 1) else 1, post_prelu=int(post_prelu),
        out=out.data_ptr(), out_pitch=_pitch4(out), out_coff=0, P=P, C=c, dtype=_DT[x.dtype], gate_a=_ptr(gate_a), gate_d=_ptr(gate_d)), _stream())
    return out


class PReLUFn(torch.autograd.Function):
    """nn.PReLU (one shared slope or one per channel) on an NHWC tensor: srk_chan_apply forward, gate + slope-gradient
    reduction backward (models/srresnet.py:14,20,27; models/ddbpn.py:33,42-53,82-86)."""

    @staticmethod
    def forward(ctx, x, weight):
        x = x.contiguous()
        ctx.save_for_backward(x, weight)
        ctx.wparam = weight
        return chan_apply(x, slope=weight, post_prelu=True)

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        want_x, want_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if x.numel() == 0:
            return (torch.empty_like(g) if want_x else None), (torch.zeros_like(weight) if want_w else None)
        if not want_w:
            return chan_apply(g, z=x, slope=weight), None
        # partial sums over x <= 0 of x * g, summed (over the channels too for a single slope) by one small launch; the pass that
        # takes them writes the input gradient g * (x > 0 ? 1 : slope) from the same read of x and g
        c = x.shape[-1]
        gate = gx = None
        if want_x:
            sl = _f32c(weight)
            if sl.numel() not in (1, c):
                sl = torch.nn.functional.pad(sl, (0, c - sl.numel()))
            gx = torch.empty_like(g)
            gate = (gx, sl)
        one = weight.numel() == 1
        slot = _grad_slot(ctx.wparam, tuple(weight.shape))      # an existing fp32 .grad: the finalize step adds into it
        acc = slot[1] if (slot is not None and slot[0] == "acc") else None
        s = chan_reduce(x, g, 2, None, 4, 1, total=one, creal=None if one else weight.numel(), dgamma_acc=acc, gate=gate)[0]
        gw = None if acc is not None else (s[:1] if one else s[:weight.numel()])
        return gx, gw


def prelu(x, weight):
    return PReLUFn.apply(x, weight)


class BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d on an NHWC tensor, training (batch statistics, running buffers updated like torch) or eval mode,
    optionally fused with a residual add: out = gamma*(x - mean)*invstd + beta (+ res).
    Reference: the `norm` of `ResBlock` / `BasicBlock` (models/common.py:33-56,97-98) in SRResNet (srresnet.py:16-21).
    Statistics: srk_chan_stats (fp32 sums); apply and backward: srk_chan_apply; [C]-sized vector math stays in torch."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, res, nbt=None, link=None):
        x = x.contiguous()
        ctx.params = (weight, bias)
        ctx.link = link
        c = weight.numel()
        cp = x.shape[-1]
        M = x.numel() // cp
        if training and M > 0:
            # ONE pass: sums of the values shifted by the tensor's first pixel K (E[x^2] - mean^2 cancels in fp32 when |mean| >> std,
            # which formula-filled / badly scaled nets do have; E[(x-K)^2] - (E[x-K])^2 with K from the data does not), and the
            # [C]-sized arithmetic (mean; variance, running buffers, invstd, scale and shift of the apply pass) in the same launch.
            w32, b32 = _f32c(weight), _f32c(bias)
            upd = running_mean is not None and running_mean.dtype == torch.float32 and running_var.dtype == torch.float32
            k0 = torch.empty(cp, dtype=torch.float32, device=x.device)
            r = chan_reduce(x, None, 0, None, 1, 5, M=M, creal=c, eps=eps, momentum=momentum, mean=k0, weight=w32, bias=b32,
                            running_mean=running_mean if upd else None, running_var=running_var if upd else None, nbt=nbt, shift_out=k0)
            mean = r[4]
            nbt = None
            invstd, gamma, a, d = r[0], r[1], r[2], r[3]
            if running_mean is not None and not upd:        # buffers in another dtype: torch arithmetic
                var = 1.0 / (invstd * invstd) - eps
                with torch.no_grad():
                    running_mean.mul_(1 - momentum).add_(mean[:c].to(running_mean.dtype), alpha=momentum)
                    running_var.mul_(1 - momentum).add_((var[:c] * (M / max(M - 1, 1))).to(running_var.dtype), alpha=momentum)
        else:
            if training:
                mean, var = torch.zeros(cp, dtype=torch.float32, device=x.device), torch.zeros(cp, dtype=torch.float32, device=x.device)
            else:
                mean = torch.nn.functional.pad(running_mean.float(), (0, cp - c))
                var = torch.nn.functional.pad(running_var.float(), (0, cp - c), value=1.0)
            invstd = torch.rsqrt(var + eps)
            gamma = torch.nn.functional.pad(weight.detach().float(), (0, cp - c))
            beta = torch.nn.functional.pad(bias.detach().float(), (0, cp - c))
            a = gamma * invstd
            d = beta - mean * a
        if nbt is not None:                 # (no batch statistics pass ran: empty batch)
            nbt.add_(1)
        out = chan_apply(x, y=res, a=a, d=d)
        ctx.save_for_backward(x, mean, invstd, gamma)
        ctx.cfg = (training, c, M, res is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        x, mean, invstd, gamma = ctx.saved_tensors
        training, c, M, has_res = ctx.cfg
        g = g.contiguous()
        if M == 0:
            z = torch.zeros(c, dtype=torch.float32, device=x.device)
            return torch.empty_like(g), z, z.clone(), None, None, None, None, None, (g if has_res else None), None, None
        mean = mean.contiguous()                                         # the sums: sum dy, sum (x - mean)*dy
        # gamma's / beta's gradients go straight into the parameters' existing fp32 .grad buffers when they have them (the finalize
        # step adds them there: what autograd's AccumulateGrad would do with one more launch each), else to autograd as tensors
        wacc, wmode = _pass_slot(ctx.params[0], (c,), x.device) if ctx.needs_input_grad[1] else (None, None)
        bacc, bmode = _pass_slot(ctx.params[1], (c,), x.device) if ctx.needs_input_grad[2] else (None, None)

        def hand(mode, t, p):           # (a second use of the same module in this pass added into the first use's tensor: nothing to hand over)
            if mode is None:
                return None
            if mode == "new+remember":
                _remember_pass_grad(p, t)
            return t
        if training:
            # dx = gamma*invstd * (dy - dbeta/M - xhat*dgamma/M),  xhat = (x - mean)*invstd
            r = chan_reduce(x, g, 1, mean, 2, 5, M=M, creal=c, mean=mean, invstd=invstd.contiguous(), gamma=gamma.contiguous(), dgamma_acc=wacc, dbeta_acc=bacc)
            dgamma, dbeta = r[0], r[1]
            gx = chan_apply(g, y=x, a=r[2], b=r[3], d=r[4])
        else:
            r = chan_reduce(x, g, 1, mean, 3, 3, M=M, creal=c, invstd=invstd.contiguous(), gamma=gamma.contiguous(), dgamma_acc=wacc, dbeta_acc=bacc)
            dgamma, dbeta = r[0], r[1]
            gx = chan_apply(g, a=r[2])
        gres = g if has_res else None
        if has_res and ctx.link is not None and ctx.link.armed:         # parked for the block's first conv (ResLink): its data-gradient launch adds it
            ctx.link.g, gres = g, None
        return gx, hand(wmode, dgamma[:c], ctx.params[0]), hand(bmode, dbeta[:c], ctx.params[1]), None, None, None, None, None, gres, None, None


class BNPReLUFn(torch.autograd.Function):
    """nn.BatchNorm2d (batch statistics) followed by nn.PReLU -- SRResNet's conv -> norm -> act (srresnet.py:16-21 through
    common.py:94-100) -- as ONE unit: forward = the statistics launch + one apply launch (the activation rides in it, the BatchNorm's
    output is never stored); backward = ONE statistics launch over (x, dy) that recomputes the BatchNorm output a x + d for the
    PReLU's gate and takes the BatchNorm's two sums AND the slope's gradient (srk_chan_stats mode 3), + one apply launch
    (dx = A dy gate + B x + D).  Five launches per layer instead of seven."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, slope, nbt):
        x = x.contiguous()
        c, cp = weight.numel(), x.shape[-1]
        M = x.numel() // cp
        w32, b32 = _f32c(weight), _f32c(bias)
        upd = running_mean is not None and running_mean.dtype == torch.float32 and running_var.dtype == torch.float32
        k0 = torch.empty(cp, dtype=torch.float32, device=x.device)
        r = chan_reduce(x, None, 0, None, 1, 5, M=M, creal=c, eps=eps, momentum=momentum, mean=k0, weight=w32, bias=b32,
                        running_mean=running_mean if upd else None, running_var=running_var if upd else None, nbt=nbt, shift_out=k0)
        invstd, gamma, a, d, mean = r[0], r[1], r[2], r[3], r[4]
        if running_mean is not None and not upd:            # buffers in another dtype: torch arithmetic
            var = 1.0 / (invstd * invstd) - eps
            with torch.no_grad():
                running_mean.mul_(1 - momentum).add_(mean[:c].to(running_mean.dtype), alpha=momentum)
                running_var.mul_(1 - momentum).add_((var[:c] * (M / max(M - 1, 1))).to(running_var.dtype), alpha=momentum)
        sl = _f32c(slope)
        if sl.numel() not in (1, cp):
            sl = torch.nn.functional.pad(sl, (0, cp - sl.numel()))
        out = chan_apply(x, a=a, d=d, slope=sl, post_prelu=True)
        ctx.save_for_backward(x, mean, invstd, gamma, a, d, sl)
        ctx.cfg = (c, M, slope.numel())
        ctx.params = (weight, bias, slope)
        return out

    @staticmethod
    def backward(ctx, g):
        x, mean, invstd, gamma, a, d, sl = ctx.saved_tensors
        c, M, ns = ctx.cfg
        g = g.contiguous()

        dev = x.device
        wacc, wmode = _pass_slot(ctx.params[0], (c,), dev) if ctx.needs_input_grad[1] else (None, None)
        bacc, bmode = _pass_slot(ctx.params[1], (c,), dev) if ctx.needs_input_grad[2] else (None, None)
        sacc, smode = _pass_slot(ctx.params[2], tuple(ctx.params[2].shape), dev) if ctx.needs_input_grad[7] else (None, None)
        r = chan_reduce(x, g, 3, mean.contiguous(), 2, 6, M=M, creal=c, mean=mean.contiguous(), invstd=invstd.contiguous(), gamma=gamma.contiguous(),
                        dgamma_acc=wacc, dbeta_acc=bacc, bn_gate=(a.contiguous(), d.contiguous(), sl), total2=ns == 1, dslope_acc=sacc)
        gx = chan_apply(g, y=x, z=x, a=r[2], b=r[3], d=r[4], slope=sl, gate_a=a.contiguous(), gate_d=d.contiguous())

        def hand(mode, t, p):
            if mode is None:
                return None
            if mode == "new+remember":
                _remember_pass_grad(p, t)
            return t
        gw = hand(wmode, r[0][:c], ctx.params[0])
        gb = hand(bmode, r[1][:c], ctx.params[1])
        gs = hand(smode, r[5][:1] if ns == 1 else r[5][:ns], ctx.params[2])
        return (gx, gw, gb, None, None, None, None, gs, None)


_BN_PRELU_FUSED = _knob("SRK_NO_BN_PRELU", "0") != "1"      # A/B knob


def batch_norm_prelu(x, bn, slope):
    """`prelu(batch_norm(x, bn), slope)` -- fused (BNPReLUFn) in training mode with batch statistics and momentum set."""
    if (_BN_PRELU_FUSED and (bn.training or bn.running_mean is None) and bn.momentum is not None and x.numel() > 0
            and slope.numel() in (1, bn.weight.numel()) and bn.weight is not None):
        nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats and bn.num_batches_tracked is not None) else None
        if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64):
            nbt.add_(1)
            nbt = None
        return BNPReLUFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, slope, nbt)
    return prelu(batch_norm(x, bn), slope)


def batch_norm(x, bn, res=None, link=None):
    """`bn`: an nn.BatchNorm2d (parameters, running buffers, training flag, momentum, eps) applied to NHWC `x`.
    link: a ResLink shared with the conv that consumes `res` (the residual's gradient then rides in that conv's data gradient)."""
    nbt = bn.num_batches_tracked if (bn.training and bn.track_running_stats and bn.num_batches_tracked is not None) else None
    if bn.momentum is not None:
        mom = bn.momentum                                   # num_batches_tracked += 1 rides in the statistics launch
    elif nbt is not None:
        nbt.add_(1)
        mom = 1.0 / float(nbt)                              # torch: cumulative moving average (a host read, as in torch)
        nbt = None
    else:
        mom = 0.0
    use_batch = bn.training or bn.running_mean is None
    if nbt is not None and not (nbt.is_cuda and nbt.dtype == torch.int64):
        nbt.add_(1)
        nbt = None
    return BatchNormFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, use_batch, mom, bn.eps, res, nbt, link if res is not None else None)
