"""Strided / transposed / large-kernel convolutions of the remaining conv models: unfold / fold, D-DBPN's projection convs
(csrc/proj.hip), the im2col forms (csrc/generic.hip).  Part of `ops` (re-exported there): split out of ops.py in round 6 by op family."""
import os

import torch

from . import _lib as L
from .ops import _ADDR_LIMIT, _DT, _f32c, _grad_slot, _grad_target, _group_for, _knob, _need_gpu, _pitch, _ptr, _stream, _tok, conv, pad16      # (ops.py imports this module at its END)
from .ops_norm import chan_apply, chan_reduce, chan_sums


# --------------------------------------------------------------------------------------------
# remaining conv models: strided / transposed / large-kernel convs, BatchNorm, PReLU
# --------------------------------------------------------------------------------------------
def _unfold_raw(x, k, stride, pad):
    n, h, w, c = x.shape
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    cols = torch.empty((n, ho, wo, k * k * c), dtype=x.dtype, device=x.device)
    L.call("srk_unfold_nhwc", L.UnfoldNhwcArgs(x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, cols=cols.data_ptr(), cols_pitch=k * k * c,
                                               N=n, H=h, W=w, C=c, K=k, stride=stride, pad=pad, Ho=ho, Wo=wo, dtype=_DT[x.dtype]), _stream())
    return cols


def _fold_raw(cols, c, k, stride, pad, ho, wo, bias=None):
    n, hi, wi, kkc = cols.shape
    assert kkc == k * k * c
    out = torch.empty((n, ho, wo, c), dtype=cols.dtype, device=cols.device)
    L.call("srk_fold_nhwc", L.FoldNhwcArgs(cols=cols.data_ptr(), cols_pitch=_pitch(cols), bias=_ptr(bias), out=out.data_ptr(), out_pitch=c, out_coff=0,
                                           N=n, Hi=hi, Wi=wi, C=c, K=k, stride=stride, pad=pad, Ho=ho, Wo=wo, dtype=_DT[cols.dtype]), _stream())
    return out


class UnfoldFn(torch.autograd.Function):
    """im2col on an NHWC tensor (srk_unfold_nhwc); backward = col2im (srk_fold_nhwc), its adjoint."""

    @staticmethod
    def forward(ctx, x, k, stride, pad):
        _need_gpu(x)
        ctx.cfg = (tuple(x.shape), k, stride, pad)
        return _unfold_raw(x.contiguous(), k, stride, pad)

    @staticmethod
    def backward(ctx, g):
        (n, h, w, c), k, stride, pad = ctx.cfg
        return _fold_raw(g.contiguous(), c, k, stride, pad, h, w), None, None, None


class FoldFn(torch.autograd.Function):
    """col2im (srk_fold_nhwc) + per-channel bias; backward = im2col of the gradient, bias gradient = its pixel sum."""

    @staticmethod
    def forward(ctx, cols, bias, c, k, stride, pad, ho, wo):
        _need_gpu(cols)
        b32 = None
        if bias is not None:
            b32 = _f32c(bias)
            if b32.numel() != c:
                b32 = torch.nn.functional.pad(b32, (0, c - b32.numel()))
        ctx.cfg = (k, stride, pad, None if bias is None else bias.numel())
        return _fold_raw(cols.contiguous(), c, k, stride, pad, ho, wo, b32)

    @staticmethod
    def backward(ctx, g):
        k, stride, pad, nb = ctx.cfg
        g = g.contiguous()
        gb = None
        if nb is not None and ctx.needs_input_grad[1]:
            gb = chan_sums(g)[0][:nb]
        return _unfold_raw(g, k, stride, pad), gb, None, None, None, None, None, None




_LK_OFF = _knob("SRK_NO_LK", "0") == "1"        # A/B knob: large kernels through im2col as in round 2


def _nhwc_view(x):
    """`x` itself when it is a dense NHWC tensor or a channel-slice view of one (the kernels take a pixel pitch), else a copy."""
    try:
        if x.data_ptr() % 16 == 0 and _pitch(x) % 8 == 0:
            return x
    except AssertionError:
        pass
    return x.contiguous()


_PROJ_OFF = _knob("SRK_NO_PROJ", "0") == "1"    # A/B knob: D-DBPN's projections through im2col / col2im as in round 2


def proj_ok(x, w, stride, pad, up):
    """Whether a D-DBPN projection conv (ddbpn.py:10-24) on NHWC `x` runs on the direct kernels (csrc/proj.hip): scale 4
    (kernel 8, stride 4, padding 2), 32 channels on both sides, 16-bit storage."""
    if _PROJ_OFF or x.dtype not in (torch.bfloat16, torch.float16) or x.numel() == 0:
        return False
    if tuple(w.shape) != (32, 32, 8, 8) or stride != 4 or pad != 2 or x.shape[3] != 32:
        return False
    if not up and (x.shape[1] % 4 or x.shape[2] % 4):
        return False
    return x.numel() * (16 if up else 1) * 2 < _ADDR_LIMIT


def _proj_launch(x, wpk_half, bias, up, slope=None, want_pre=False):
    """One srk_proj_up / srk_proj_down launch; with `slope` (fp32 [1] or [32]) the following nn.PReLU rides in the epilogue:
    returns (activation, stored conv output or None)."""
    n, h, wd, _ = x.shape
    lh, lw = (h, wd) if up else (h // 4, wd // 4)
    shape = (n, 4 * lh, 4 * lw, 32) if up else (n, lh, lw, 32)
    out = torch.empty(shape, dtype=x.dtype, device=x.device)
    pre = torch.empty(shape, dtype=x.dtype, device=x.device) if (slope is not None and want_pre) else None
    L.call("srk_proj_up" if up else "srk_proj_down",
           L.ProjArgs(x=x.data_ptr(), x_pitch=_pitch(x), out=out.data_ptr(), out_pitch=32, wpk=wpk_half.data_ptr(), bias=_ptr(bias),
                      N=n, H=lh, W=lw, dtype=_DT[x.dtype], slope=_ptr(slope), slope_stride=0 if (slope is None or slope.numel() == 1) else 1,
                      pre=_ptr(pre), pre_pitch=32), _stream())
    return out, pre


class ProjFn(torch.autograd.Function):
    """nn.Conv2d / nn.ConvTranspose2d(32, 32, 8, stride=4, padding=2) [+ the nn.PReLU(32) behind it] on an NHWC 16-bit tensor
    (ddbpn.py:10-24, 42-53): forward, data gradient and weight gradient on the direct kernels of csrc/proj.hip.  Both weight
    layouts read as [c_low][c_high][ky][kx] (Conv2d: [out][in], ConvTranspose2d: [in][out]), so `up` alone tells the directions
    apart.  With `slope` the activation is applied in the conv's epilogue (the stored conv output is kept for the backward:
    PReLU's input gradient and slope gradient come from one pass over it, srk_chan_stats mode 2 with gate_out)."""

    @staticmethod
    def forward(ctx, x, w, b, slope, up):
        _need_gpu(x)
        x = _nhwc_view(x)
        half = L.load().srk_proj_pack_bytes() // 2
        # a model's forward window (forward_scope) packs the projection weights it has seen before in ONE launch; a first use, a
        # weight that is not a plain fp32 parameter, or a call outside any window packs here
        group = _group_for(None) if (isinstance(w, torch.nn.Parameter) and w.dtype == torch.float32 and w.is_contiguous()) else None
        key = (id(w), x.dtype)
        wpk = group.lookup_proj(key) if group is not None else None
        ctx.pg = _tok() if wpk is not None else None
        if wpk is None or wpk.device != x.device:
            wpk = torch.empty(2 * half, dtype=torch.uint8, device=x.device)
            L.check(L.load().srk_proj_pack(_f32c(w).data_ptr(), wpk.data_ptr(), _DT[x.dtype], _stream()), "srk_proj_pack")
            if group is not None:
                group.add_proj(key, w, wpk)
                ctx.pg = _tok()
        sl = None if slope is None else _f32c(slope)
        need_pre = sl is not None and (ctx.needs_input_grad[0] or ctx.needs_input_grad[1] or ctx.needs_input_grad[3])
        out, pre = _proj_launch(x, wpk[half:] if up else wpk[:half], None if b is None else _f32c(b), up, sl, need_pre)
        ctx.save_for_backward(x, wpk, pre, sl)
        ctx.up, ctx.half = bool(up), half
        ctx.wparam, ctx.bparam, ctx.sparam = w, b, slope
        return out

    @staticmethod
    def backward(ctx, g):
        x, wpk, pre, sl = ctx.saved_tensors
        up, half = ctx.up, ctx.half
        if ctx.pg is not None and _group_for(ctx.pg) is None:
            # the group's buffer was re-packed by a later forward window (from possibly updated weights): pack the weights again, here
            wpk = torch.empty(2 * half, dtype=torch.uint8, device=x.device)
            L.check(L.load().srk_proj_pack(_f32c(ctx.wparam).data_ptr(), wpk.data_ptr(), _DT[x.dtype], _stream()), "srk_proj_pack")
        g = _nhwc_view(g)               # (a slice of a SliceBuffer's gradient buffer is read with its pitch: no copy)
        gs = None
        if sl is not None:                       # through the PReLU first: g <- g * (pre > 0 ? 1 : slope), slope gradient on the side
            if ctx.needs_input_grad[3]:
                one = sl.numel() == 1
                slot = _grad_slot(ctx.sparam, tuple(ctx.sparam.shape))
                acc = slot[1] if (slot is not None and slot[0] == "acc") else None
                gp = torch.empty_like(g)
                s = chan_reduce(pre, g, 2, None, 4, 1, total=one, creal=None if one else sl.numel(), dgamma_acc=acc, gate=(gp, sl))[0]
                gs = None if acc is not None else (s[:1] if one else s[:sl.numel()])
                g = gp
            else:
                g = chan_apply(g, z=pre, slope=sl)
        gx = _proj_launch(g, wpk[:half] if up else wpk[half:], None, not up)[0] if ctx.needs_input_grad[0] else None
        gw = gb = None
        want_b = ctx.bparam is not None and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            xh, gl = (g, x) if up else (x, g)
            n, lh, lw, _ = gl.shape

            def slot_of(p, shape):          # existing fp32 .grad: added into; GradSync's flat-buffer slice: written there; else fresh
                sl_ = _grad_slot(p, shape)
                acc_ = sl_[1] if (sl_ is not None and sl_[0] == "acc") else None
                t = acc_ if acc_ is not None else _grad_target(p, shape, x.device)
                return (t if t is not None else torch.empty(shape, dtype=torch.float32, device=x.device)), acc_ is not None
            dw, wacc = slot_of(ctx.wparam, (32, 32, 8, 8))
            db, bacc = slot_of(ctx.bparam, (32,)) if want_b else (None, False)
            scratch = torch.empty(L.load().srk_proj_wgrad_scratch_floats(n, lh, lw), dtype=torch.float32, device=x.device)
            L.call("srk_proj_wgrad", L.ProjWgradArgs(xh=xh.data_ptr(), xh_pitch=_pitch(xh), g=gl.data_ptr(), g_pitch=_pitch(gl),
                                                     scratch=scratch.data_ptr(), dw=dw.data_ptr(), accumulate=int(wacc),
                                                     N=n, H=lh, W=lw, dtype=_DT[x.dtype], db=_ptr(db), bias_side=2 if up else 1,
                                                     db_accumulate=int(bacc)), _stream())
            gw = None if wacc else dw
            gb = None if (bacc or not want_b) else db
        elif want_b:
            gb = chan_sums(g)[0][:32]
        return gx, gw, gb, gs, None


def proj_prelu(x, w, b, slope, *, up):
    """D-DBPN's [projection conv, PReLU] pair (ddbpn.py:42-53) on NHWC `x` as ONE forward launch (shapes: see proj_ok)."""
    return ProjFn.apply(x, w, b, slope, bool(up))


def conv_general(x, w, b, *, stride=1, pad=0):
    """nn.Conv2d with any square kernel / stride / zero padding on NHWC `x`: im2col (srk_unfold_nhwc) + the 1x1 MFMA conv
    with the OIHW weight presented as a [Cout][K*K*Cin] matrix in (kh, kw, ci) order.  The permute / reshape of the
    parameter is a view-level torch op, so its gradient flows back to the OIHW parameter through autograd."""
    cout, cin, k, _ = w.shape
    cp = x.shape[-1]
    if proj_ok(x, w, stride, pad, False):             # D-DBPN's down-projection at scale 4: direct kernels (csrc/proj.hip)
        return ProjFn.apply(x, w, b, None, False)
    if (stride == 1 and pad == k // 2 and k in (5, 7, 9) and cin == cp == 64 and cout <= 16 and x.dtype in (torch.bfloat16, torch.float16)
            and x.numel() * 2 < _ADDR_LIMIT and not _LK_OFF):
        # SRResNet's 9x9 tail conv (srresnet.py:29): the direct large-kernel kernels (csrc/conv_lk.hip), no column tensor
        return conv(x, w, b)
    if cp != cin:                                   # zero-padded storage channels: pad the weight's input channels too
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, cp - cin))
    wm = w.permute(0, 2, 3, 1).reshape(cout, k * k * cp, 1, 1)
    cols = UnfoldFn.apply(x, k, stride, pad)
    return conv(cols, wm, b)


def conv_transpose_general(x, w, b, *, stride=1, pad=0):
    """nn.ConvTranspose2d (weight [Cin][Cout][K][K]) on NHWC `x`: 1x1 MFMA conv to K*K*Cout channels + col2im gather
    (srk_fold_nhwc) with the bias added once per output element."""
    cin, cout, k, _ = w.shape
    n, h, wd, cp = x.shape
    if proj_ok(x, w, stride, pad, True):              # D-DBPN's up-projection at scale 4: direct kernels (csrc/proj.hip)
        return ProjFn.apply(x, w, b, None, True)
    coutp = pad16(cout)
    if cp != cin:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, 0, 0, cp - cin))
    if coutp != cout:
        w = torch.nn.functional.pad(w, (0, 0, 0, 0, 0, coutp - cout))
    wm = w.permute(2, 3, 1, 0).reshape(k * k * coutp, cp, 1, 1)
    cols = conv(x, wm, None)
    ho, wo = (h - 1) * stride - 2 * pad + k, (wd - 1) * stride - 2 * pad + k
    return FoldFn.apply(cols, b, coutp, k, stride, pad, ho, wo)
