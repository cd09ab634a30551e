"""Host side of the HIP hot path: thin launch wrappers + block-level `torch.autograd.Function`s.

Activations between ops are NHWC tensors `[N, H, W, Cp]` in the compute dtype
(bf16 / fp16 / fp32), Cp = channels padded to a multiple of 16 with zero padding
channels; NCHW fp32 exists only at the model boundary (`head_conv` in,
`tail_conv` out), exactly where the reference's `forward(x)` contract is
(models/srmodel.py:156-171).  Parameters of record stay OIHW fp32
(state_dict-compatible); packed MFMA layouts are shadow copies
cached per parameter version.

Nothing here falls back to PyTorch arithmetic: every op raises if the tensors are
not on a GPU or the HIP library is missing.
"""
import contextlib
import os

import torch

from . import _lib as L

_DT = {torch.bfloat16: L.SRK_BF16, torch.float16: L.SRK_F16, torch.float32: L.SRK_F32}


def _knob(name, default="0"):
    """A/B switches of tools/ (SRK_NO_PAIR, SRK_NO_HR_COLLAPSE, ...): read ONLY under SRK_DEBUG=1, so that a stray variable in a user's
    environment cannot silently route the product path through a slower diagnostic form."""
    return os.environ.get(name, default) if os.environ.get("SRK_DEBUG") == "1" else default


def pad16(c):
    return (int(c) + 15) // 16 * 16


def _roundup(a, b):
    return (a + b - 1) // b * b


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _need_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("the sr_amd HIP ops run on an MI355X ('cuda') device only; there is no CPU fallback "
                           "(use oracle/ for a CPU reference in tests)")


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _pitch(t):
    """Pixel pitch (elements) of an NHWC tensor / channel-slice view; validates the layout."""
    n, h, w, c = t.shape
    p = t.stride(2) if w > 1 else (t.stride(1) if h > 1 else (t.stride(0) if n > 1 else c))
    if w > 1 and h > 1:
        assert t.stride(1) == w * p, "NHWC view must be dense in W"
    if h > 1 and n > 1:
        assert t.stride(0) == h * w * p, "NHWC view must be dense in H"
    assert c == 1 or t.stride(3) == 1, "channels must be innermost"
    return p


# --------------------------------------------------------------------------------------------
# weight packing (cached on the parameter object, keyed by its in-place version counter)
# --------------------------------------------------------------------------------------------
class Packed:
    __slots__ = ("wpk", "bias", "KinP", "CoutP", "k", "ps_r", "cr")


_PACK_CACHE_ENABLED = False


class pack_cache:
    """Context manager: reuse packed weights across calls while the parameters are FROZEN (inference loops).

    Off by default: an optimizer may update parameters without touching their autograd version counter
    (torch's fused Adam does), so a version-keyed cache cannot be trusted while training -- every forward /
    backward repacks (one small kernel per conv)."""

    def __init__(self, enabled=True):
        self.enabled, self.prev = enabled, None

    def __enter__(self):
        global _PACK_CACHE_ENABLED
        self.prev, _PACK_CACHE_ENABLED = _PACK_CACHE_ENABLED, self.enabled
        return self

    def __exit__(self, *exc):
        global _PACK_CACHE_ENABLED
        _PACK_CACHE_ENABLED = self.prev


_PACK_TILED = _knob("SRK_NO_PACK_TILED", "0") != "1"      # A/B knob: the strided-read grouped pack launch


class PackGroup:
    """The packed shadow weights of ONE model, re-packed by a single kernel launch per step.

    `forward_scope(group)` (entered at the top of a model's forward) launches `srk_pack_conv_weights_group` over a
    device-side table of every (parameter, layout) pair the model has used so far -- forward and dgrad layouts --
    so the ~2 pack launches per conv per training step collapse into one.  Entries are discovered on the first
    step (packed individually then).  Buffers are persistent, so the launch is hipGraph-capturable.

    Validity: packed buffers are served ONLY (a) inside the forward window that refreshed them and (b) to the
    backward of a graph built in that window (the Functions carry the (group, epoch) token they were built under and
    a newer refresh invalidates it).  Anything else -- a block called on its own after the model's forward, a sub-module
    called directly, another model's parameters -- packs per call from the current parameter values: an optimizer step
    between two forwards can never be served stale weights (torch's fused Adam does not bump `_version`)."""

    def __init__(self):
        self.entries = {}       # key -> [PackArgs, Packed, w, b]
        self.pw_entries = {}    # key -> [PwPackArgs, PwPacked, (w1, b1, w2, b2)]  (fused pointwise pairs, csrc/pw_chain.hip)
        self.pw_table = None
        self.proj_entries = {}  # (id(w), dtype) -> [w, packed buffer]  (D-DBPN's projection convs, csrc/proj.hip)
        self.proj_tables = {}   # dtype -> [device table, addresses it was built from]
        self.tiles, self.total_tiles = None, 0
        self.pw_dirty = False
        self.table = None
        self.dirty = False
        self.epoch = 0          # number of refreshes so far
        self.open = False       # inside the forward window of the latest refresh

    def lookup(self, key):
        e = self.entries.get(key)
        return e[1] if e is not None else None

    @staticmethod
    def _held(t):
        """What an entry keeps of a source tensor: the tensor itself for a leaf (its address is re-read every refresh), a
        DETACHED alias for anything else.  A weight-normed conv's effective weight carries the grad_fn of the step that made
        it; held here, that node would keep the AccumulateGrad nodes of an eager step (created on the eager stream) alive
        into a later hipGraph capture, whose backward then ties the eager stream into the capture and hipStreamEndCapture
        crashes."""
        return t if (t is None or t.grad_fn is None) else t.detach()

    def add(self, key, args, packed, w, b):
        self.entries[key] = [args, packed, self._held(w), self._held(b)]
        self.dirty = True

    def lookup_pw(self, key):
        e = self.pw_entries.get(key)
        return e[1] if e is not None else None

    def add_pw(self, key, args, packed, tensors):
        self.pw_entries[key] = [args, packed, tuple(self._held(t) for t in tensors)]
        self.pw_dirty = True

    def lookup_proj(self, key):
        e = self.proj_entries.get(key)
        return e[1] if e is not None else None

    def add_proj(self, key, w, wpk):
        self.proj_entries[key] = [w, wpk]

    def _refresh_proj(self):
        """All projection weights of one storage dtype -> fragment order in ONE launch (was one 5 us launch per conv and step)."""
        import ctypes as C
        by_dt = {}
        for (_, dt), (w, wpk) in self.proj_entries.items():
            by_dt.setdefault(dt, []).append((w, wpk))
        for dt, ents in by_dt.items():
            addrs = tuple((w.data_ptr(), wpk.data_ptr()) for w, wpk in ents)
            tb = self.proj_tables.get(dt)
            if tb is None or tb[1] != addrs:
                host = (L.ProjPackJob * len(ents))()
                for i, (wa, pa) in enumerate(addrs):
                    host[i].w4, host[i].wpk = wa, pa
                raw = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8)
                dev = ents[0][1].device
                if tb is None or tb[0].numel() < raw.numel():       # the table keeps its address while it fits (a captured step names it)
                    tb = [torch.empty(max(2 * raw.numel(), 1024), dtype=torch.uint8, device=dev), None]
                tb[0][:raw.numel()].copy_(raw)
                tb[1] = addrs
                self.proj_tables[dt] = tb
            L.check(L.load().srk_proj_pack_group(tb[0].data_ptr(), len(ents), _DT[dt], _stream()), "srk_proj_pack_group")

    def _refresh_pw(self):
        for e in self.pw_entries.values():
            a, _, (w1, b1, w2, b2) = e
            ptrs = (w1.data_ptr(), _ptr(b1), w2.data_ptr(), _ptr(b2))
            if (a.w1, a.b1 or 0, a.w2, a.b2 or 0) != ptrs:
                a.w1, a.b1, a.w2, a.b2 = ptrs
                self.pw_dirty = True
        if self.pw_dirty:
            arr = (L.PwPackArgs * len(self.pw_entries))(*[e[0] for e in self.pw_entries.values()])
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev = next(iter(self.pw_entries.values()))[2][0].device
            if self.pw_table is None or self.pw_table.device != dev or self.pw_table.numel() < raw.numel():
                self.pw_table = torch.empty(max(2 * raw.numel(), 4096), dtype=torch.uint8, device=dev)
            self.pw_table[:raw.numel()].copy_(raw)
            self.pw_dirty = False
        L.check(L.load().srk_pw_pack_group(self.pw_table.data_ptr(), len(self.pw_entries), _stream()), "srk_pw_pack_group")

    def refresh(self):
        self.epoch += 1
        if self.pw_entries:
            self._refresh_pw()
        if self.proj_entries:
            self._refresh_proj()
        if not self.entries:
            return
        for e in self.entries.values():          # parameters moved / re-allocated since the table was built?
            a, _, w, b = e
            bp = 0 if b is None else b.data_ptr()
            if a.w != w.data_ptr() or (a.bias or 0) != bp:
                a.w, a.bias = w.data_ptr(), bp
                self.dirty = True
        if self.dirty:
            import ctypes as C
            n = len(self.entries)
            arr = (L.PackArgs * n)(*[e[0] for e in self.entries.values()])
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            dev = next(iter(self.entries.values()))[2].device
            # the table keeps its ADDRESS while it fits (a captured step points at it): grown in place, with head room
            if self.table is None or self.table.device != dev or self.table.numel() < raw.numel():
                self.table = torch.empty(max(2 * raw.numel(), 4096), dtype=torch.uint8, device=dev)
            self.table[:raw.numel()].copy_(raw)
            # blocks of the launch -> (entry, tile): prefix sums of the entries' tile counts, next to the table
            tb = (C.c_int * (n + 1))()
            L.check(L.load().srk_pack_group_tiles(C.addressof(arr), n, C.addressof(tb)), "srk_pack_group_tiles")
            if self.tiles is None or self.tiles.device != dev or self.tiles.numel() < n + 1:
                self.tiles = torch.empty(max(2 * (n + 1), 1024), dtype=torch.int32, device=dev)
            self.tiles[:n + 1].copy_(torch.tensor(list(tb), dtype=torch.int32))
            self.total_tiles = int(tb[n])
            self.dirty = False
        if _PACK_TILED:
            L.check(L.load().srk_pack_conv_weights_group_tiled(self.table.data_ptr(), self.tiles.data_ptr(), len(self.entries), self.total_tiles, _stream()),
                    "srk_pack_conv_weights_group_tiled")
        else:
            L.check(L.load().srk_pack_conv_weights_group(self.table.data_ptr(), len(self.entries), _stream()), "srk_pack_conv_weights_group")


import threading
import weakref
_TLS = threading.local()


class forward_scope:
    """`with forward_scope(group):` around a model's forward: refreshes `group` (one launch) and makes it the group that
    serves / registers packed weights until the block exits.  `group=None`: no grouping (every conv packs per use)."""

    def __init__(self, group):
        self.group = group

    def __enter__(self):
        self.prev = getattr(_TLS, "group", None)
        _TLS.group = self.group
        if self.group is not None:
            self.group.refresh()
            self.group.open = True
        return self

    def __exit__(self, *exc):
        if self.group is not None:
            self.group.open = False
        _TLS.group = self.prev


def _tok():
    """(group, epoch) token of the forward window a Function is being built in, or None."""
    g = getattr(_TLS, "group", None)
    return (weakref.ref(g), g.epoch) if (g is not None and g.open) else None


def _group_for(token):
    """The PackGroup allowed to serve this call: the open forward window's, or the one named by a still-current token."""
    if token is not None:
        g = token[0]()
        return g if (g is not None and g.epoch == token[1]) else None
    g = getattr(_TLS, "group", None)
    return g if (g is not None and g.open) else None


def pack_conv(w, b, dtype, *, dgrad=False, ps_r=0, cache=True, as_1x1=False, token=None):
    """OIHW fp32 `w` (+ bias) -> packed shadow layout for srk_conv2d (forward or dgrad).
    as_1x1: present the OIHW weight as the 1x1 conv over Cin*KH*KW unfolded channels (head / skip convs).
    token: the (group, epoch) a Function's forward was built under (backward calls pass it: see PackGroup)."""
    _need_gpu(w)
    # a weight-normed conv's effective weight (WeightNormGroup) is a non-leaf tensor in a buffer with a STABLE address: it joins
    # the grouped pack launch under the identity of its `weight_v` parameter
    kobj = None if isinstance(w, torch.nn.Parameter) else w.__dict__.get("_srk_pack_key")
    is_param = isinstance(w, torch.nn.Parameter) or kobj is not None
    key = (id(w) if kobj is None else kobj, dtype, bool(dgrad), int(ps_r), bool(as_1x1))
    group = _group_for(token) if is_param else None
    if group is not None:
        hit = group.lookup(key)
        if hit is not None:
            return hit
    ver = (w._version, -1 if b is None else b._version, w.data_ptr())
    store = None
    if cache and _PACK_CACHE_ENABLED and is_param and kobj is None:
        store = w.__dict__.setdefault("_srk_pack", {})
        hit = store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
    cout, cin, kh, kw = w.shape
    assert kh == kw
    if as_1x1:
        cin, kh, kw = cin * kh * kw, 1, 1
    p = Packed()
    p.k, p.ps_r = kh, int(ps_r)
    if not dgrad:
        p.KinP, p.CoutP = pad16(cin), _roundup(cout, L.conv_tile(cout))
    else:
        p.KinP, p.CoutP = pad16(cout), _roundup(cin, L.conv_tile(cin))
    wf = w.detach()
    if wf.dtype != torch.float32 or not wf.is_contiguous():
        wf = wf.float().contiguous()
    # large kernel with few output channels (SRResNet's 9x9 tail, 64 -> 3): the forward kernel wants (kw, co) pairs on the MFMA rows;
    # that layout (kh * 2048 elements) rides behind the standard one
    rows = (not dgrad and not as_1x1 and kh in (5, 7, 9) and cin == 64 and cout <= 4 and cout * kw <= 32 and int(ps_r) <= 1
            and dtype in (torch.bfloat16, torch.float16) and p.CoutP == 32)
    p.cr = int(cout) if rows else 0
    p.wpk = torch.empty(kh * kw * p.KinP * p.CoutP + (kh * 2048 if rows else 0), dtype=dtype, device=w.device)
    p.bias = None
    bf = None
    if not dgrad:
        p.bias = torch.empty(p.CoutP, dtype=torch.float32, device=w.device)
        if b is not None:
            bf = b.detach()
            if bf.dtype != torch.float32 or not bf.is_contiguous():
                bf = bf.float().contiguous()
    a = L.PackArgs(w=wf.data_ptr(), bias=_ptr(bf), wpk=p.wpk.data_ptr(), bias_pk=_ptr(p.bias),
                   Cout=cout, Cin=cin, KH=kh, KW=kw, KinP=p.KinP, CoutP=p.CoutP,
                   dgrad=int(dgrad), ps_r=int(ps_r), dtype=_DT[dtype], rows_layout=int(rows))
    L.call("srk_pack_conv_weights", a, _stream())
    if store is not None:
        store[key] = (ver, p)
    if group is not None and wf.data_ptr() == w.data_ptr() and (bf is None or bf.data_ptr() == b.data_ptr()):
        group.add(key, a, p, w, b if not dgrad else None)
    return p


# --------------------------------------------------------------------------------------------
# raw launch wrappers
# --------------------------------------------------------------------------------------------
_ADDR_LIMIT = 0x7fff0000      # bytes one launch of the 32-bit-offset (buffer instruction) kernels can address per tensor


def _batch_chunks(n, *tensors):
    """Number of batch chunks so that every tensor's chunk stays below _ADDR_LIMIT bytes (1 for ordinary sizes)."""
    worst = 0
    for t in tensors:
        if t is not None and t.dim() > 0 and t.shape[0] > 0:
            worst = max(worst, t.shape[0] * t.stride(0) * t.element_size())
    if worst < _ADDR_LIMIT or n <= 1:
        return 1
    return min(n, -(-worst // (_ADDR_LIMIT // 2)))


_SIGN_BITS = _knob("SRK_NO_SIGN_BITS") != "1"       # A/B knob: ReLU backward masks re-read the activation instead of its sign bits


def conv_raw(x, pk, *, N, H, W, Cin, Cout, out, out_mode=L.OUT_NHWC, ps_r=0, relu=False, scale=1.0,
             res=None, mask=None, mask_from=0, post_add=None, x_ps=0, use_bias=True, relu_bits=None, mask_bits=None):
    """One srk_conv2d launch.  `x`, `out`, `res`, `mask` are NHWC tensors or channel-slice views
    (planar mode: `out`/`res` are NCHW fp32).  (N,H,W) are the conv-space dims.
    relu_bits: "want" -> the launch also writes the ReLU sign bits of its output ([N*H*W, 2] int32, include/srk.h) when the kernel that
    serves this shape can (returned as `out._srk_bits`, else None); mask_bits: such a tensor, used INSTEAD of `mask` (the caller
    passes both: `mask` is the fallback when the kernel cannot take bits)."""
    _need_gpu(x)
    if relu_bits is not None or mask_bits is not None:
        # (a pixel-shuffled or post_add launch is not a bits launch: the bit words are indexed by the conv's own output pixel)
        ok = _SIGN_BITS and _batch_chunks(N, x, out, res, mask) == 1 and mask_from == 0 and out_mode == L.OUT_NHWC and int(ps_r) <= 1 and post_add is None
        bits_t = None
        if ok:
            bits_t = mask_bits if mask_bits is not None else torch.empty((N * H * W, 2), dtype=torch.int32, device=x.device)
            probe = L.ConvArgs(x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, x_ps=int(x_ps), N=N, H=H, W=W, Cin=Cin, wpk=pk.wpk.data_ptr(), bias=0,
                               CoutP=pk.CoutP, Cout=Cout, KH=pk.k, KW=pk.k, relu=int(relu), scale=float(scale), res=_ptr(res),
                               res_pitch=0 if res is None else _pitch(res), res_coff=0, mask=0, mask_pitch=0, mask_coff=0, mask_from=0,
                               out=out.data_ptr(), out_pitch=_pitch(out), out_coff=0, out_mode=out_mode, ps_r=int(ps_r), post_add=0,
                               dtype=_DT[x.dtype], cout_real=0, relu_bits=bits_t.data_ptr() if relu_bits is not None else 0,
                               mask_bits=bits_t.data_ptr() if mask_bits is not None else 0)
            ok = bool(L.load().srk_conv_bits_ok(probe))
        if not ok:
            conv_raw(x, pk, N=N, H=H, W=W, Cin=Cin, Cout=Cout, out=out, out_mode=out_mode, ps_r=ps_r, relu=relu, scale=scale, res=res, mask=mask,
                     mask_from=mask_from, post_add=post_add, x_ps=x_ps, use_bias=use_bias)
            out.__dict__["_srk_bits"] = None
            return out
        a = probe
        a.bias = _ptr(pk.bias) if use_bias else 0
        a.cout_real = getattr(pk, "cr", 0) or 0
        L.call("srk_conv2d", a, _stream())
        out.__dict__["_srk_bits"] = bits_t if relu_bits is not None else None
        return out
    nck = _batch_chunks(N, x, out, res, mask)
    if nck > 1:         # a tensor of 2 GiB or more: the persistent kernels address 31 bits -> several launches over the batch
        step = -(-N // nck)
        for n0 in range(0, N, step):
            n1 = min(N, n0 + step)
            conv_raw(x[n0:n1], pk, N=n1 - n0, H=H, W=W, Cin=Cin, Cout=Cout, out=out[n0:n1], out_mode=out_mode, ps_r=ps_r,
                     relu=relu, scale=scale, res=None if res is None else res[n0:n1], mask=None if mask is None else mask[n0:n1],
                     mask_from=mask_from, post_add=post_add, x_ps=x_ps, use_bias=use_bias)
        return out
    dt = x.dtype
    planar = out_mode == L.OUT_PLANAR
    a = L.ConvArgs(
        x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, x_ps=int(x_ps), N=N, H=H, W=W, Cin=Cin,
        wpk=pk.wpk.data_ptr(), bias=_ptr(pk.bias) if use_bias else 0, CoutP=pk.CoutP, Cout=Cout, KH=pk.k, KW=pk.k,
        relu=int(relu), scale=float(scale),
        res=_ptr(res), res_pitch=0 if (res is None or planar) else _pitch(res), res_coff=0,
        mask=_ptr(mask), mask_pitch=0 if mask is None else _pitch(mask), mask_coff=0, mask_from=int(mask_from),
        out=out.data_ptr(), out_pitch=0 if planar else _pitch(out), out_coff=0, out_mode=out_mode, ps_r=int(ps_r),
        post_add=_ptr(post_add), dtype=_DT[dt], cout_real=getattr(pk, "cr", 0) or 0, relu_bits=0, mask_bits=0)
    L.call("srk_conv2d", a, _stream())
    return out


_PAIR_OFF = _knob("SRK_NO_PAIR", "0") == "1"
_CA_UNFUSED = _knob("SRK_CA_UNFUSED", "0") == "1"      # A/B knob: the CALayer backward as its own launch
_PAIR_MAX_TILES = int(_knob("SRK_PAIR_MAX_TILES", "0"))


def pair_ok(x, w1, w2):
    """Whether two chained 3x3 64->64 convs on `x` (NHWC) run as ONE srk_conv_pair launch: 16-bit storage, 64 channels and
    so few tiles (about two 14x14 output tiles per CU or less: the reference's 16 patches of 48x48 are 256) that launches,
    not MFMAs, set the time.  Larger batches keep the weight-stationary kernel, whose prologue amortises over many tiles."""
    if _PAIR_OFF or x.dtype not in (torch.bfloat16, torch.float16) or x.shape[3] != 64:
        return False
    if tuple(w1.shape) != (64, 64, 3, 3) or tuple(w2.shape) != (64, 64, 3, 3):
        return False
    n, h, wd, _ = x.shape
    if n == 0 or x.numel() * 2 >= _ADDR_LIMIT:
        return False
    lim = _PAIR_MAX_TILES or 2 * L.load().srk_device_cus()
    return L.load().srk_conv_pair_tiles(n, h, wd) <= lim


def conv_pair_raw(x, pk1, pk2, *, out, relu_mid=False, scale_mid=1.0, mask=None, mid=None, scale_out=1.0, res=None,
                  use_bias=True, pool=None, pool_aux=None, ca_bwd=None, ca_fwd=None, xo=None):
    """One srk_conv_pair launch: out = (conv(epi(conv(x, pk1)), pk2)) * scale_out + res (include/srk.h)."""
    _need_gpu(x)
    n, h, wd, _ = x.shape
    from_x = res is not None and res.data_ptr() == x.data_ptr() and _pitch(res) == _pitch(x) and ca_bwd is None and ca_fwd is None
    # ca_bwd: gsum, sums, s, z, w1, w2, slots -- the CALayer backward on the way in
    # ca_fwd: x2, sums, w1, b1, w2, b2, s_out, z_out -- the previous block's CALayer forward (t * s + x2) on the way in
    ca = ca_bwd or ca_fwd or {}
    x2 = ca.get("x2")
    a = L.ConvPairArgs(
        x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, N=n, H=h, W=wd,
        w1=pk1.wpk.data_ptr(), b1=_ptr(pk1.bias) if use_bias else 0, w2=pk2.wpk.data_ptr(), b2=_ptr(pk2.bias) if use_bias else 0,
        relu_mid=int(relu_mid), scale_mid=float(scale_mid),
        mask=_ptr(mask), mask_pitch=0 if mask is None else _pitch(mask), mask_coff=0,
        mid=_ptr(mid), mid_pitch=0 if mid is None else _pitch(mid), mid_coff=0,
        scale_out=float(scale_out),
        res=_ptr(res), res_pitch=0 if res is None else _pitch(res), res_coff=0, res_from_x=int(from_x),
        out=out.data_ptr(), out_pitch=_pitch(out), out_coff=0, dtype=_DT[x.dtype],
        pool=_ptr(pool), pool_aux=_ptr(pool_aux), pool_aux_pitch=0 if pool_aux is None else _pitch(pool_aux), pool_aux_coff=0,
        ca_mode=1 if ca_bwd is not None else (2 if ca_fwd is not None else 0), ca_cr=ca["w1"].shape[0] if ca else 0,
        ca_gsum=_ptr(ca.get("gsum")), ca_gsum_rows=ca["gsum"].shape[1] if ca_bwd else 0,
        ca_sums=_ptr(ca.get("sums")), ca_sums_rows=ca["sums"].shape[1] if ca else 0,
        ca_s=_ptr(ca.get("s")), ca_z=_ptr(ca.get("z")), ca_w1=_ptr(ca.get("w1")), ca_w2=_ptr(ca.get("w2")), ca_slots=_ptr(ca.get("slots")),
        xo=_ptr(xo), xo_pitch=0 if xo is None else _pitch(xo), xo_coff=0,
        ca_x2=_ptr(x2), ca_x2_pitch=0 if x2 is None else _pitch(x2), ca_x2_coff=0,
        ca_b1=_ptr(ca.get("b1")), ca_b2=_ptr(ca.get("b2")), ca_s_out=_ptr(ca.get("s_out")), ca_z_out=_ptr(ca.get("z_out")))
    L.call("srk_conv_pair", a, _stream())
    PAIR_LAUNCHES[a.ca_mode] += 1
    return out


def _grad_target(p, shape, dev):
    """Where the gradient of parameter `p` is to be WRITTEN, or None (a fresh tensor).  trainer.GradSync keeps all gradients in
    one flat fp32 buffer (what the bucket all-reduces run on) and names each parameter's slice here: the weight-gradient
    kernels then write the slice directly and autograd adopts a view of it as `.grad` -- no per-step pack copy.  Only while the
    parameter has no gradient yet (an existing one is accumulated into, as before)."""
    if p is None or not isinstance(p, torch.Tensor) or not p.is_leaf or p.grad is not None:
        return None
    t = p.__dict__.get("_srk_grad_target")
    if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
        return None
    # only the FIRST use of the parameter in a backward pass gets the slice: `.grad` stays None until AccumulateGrad has seen every
    # use, so a second use (a conv shared between two places of a user's model) would overwrite the first one's result in the same
    # memory and autograd would then add two aliases of it (2 g_last instead of g_1 + g_2).  Later uses return fresh tensors.
    pid = _pass_id()
    if pid >= 0:
        if p.__dict__.get("_srk_target_pass") == pid:
            return None
        p.__dict__["_srk_target_pass"] = pid
    return t.detach()           # a fresh tensor object on the same memory (AccumulateGrad adopts a gradient nobody else references)


def wgrad_raw(x, dy, *, N, H, W, Cin, Cout, k, w_shape, ps_r=0, scale=1.0, x_ps=0, dy_ps=0, want_bias=True, out_w=None, out_b=None):
    """dW (OIHW fp32) and db for a conv whose input was `x` and output gradient is `dy`.
    Cin/Cout are the padded storage channel counts of x / dy; w_shape the real OIHW shape.  out_w / out_b: write there."""
    _need_gpu(x)
    dev = x.device
    cout, cin, kh, kw = w_shape
    if N == 0:          # empty batch: zero gradients (the slab scratch would be uninitialised)
        return (torch.zeros(w_shape, dtype=torch.float32, device=dev),
                torch.zeros(cout, dtype=torch.float32, device=dev) if want_bias else None)
    nck = _batch_chunks(N, x, dy)
    if nck > 1:         # 2 GiB and more: sum the gradients of batch chunks (each chunk keeps the slab kernels)
        step = -(-N // nck)
        dw = db = None
        for n0 in range(0, N, step):
            n1 = min(N, n0 + step)
            w_, b_ = wgrad_raw(x[n0:n1], dy[n0:n1], N=n1 - n0, H=H, W=W, Cin=Cin, Cout=Cout, k=k, w_shape=w_shape, ps_r=ps_r,
                               scale=scale, x_ps=x_ps, dy_ps=dy_ps, want_bias=want_bias)
            dw = w_ if dw is None else dw.add_(w_)
            db = b_ if (db is None or b_ is None) else db.add_(b_)
        if out_w is not None:
            dw = out_w.copy_(dw)
        if out_b is not None and db is not None:
            db = out_b.copy_(db)
        return dw, db
    a = L.WgradArgs(x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, x_ps=int(x_ps),
                    dy=dy.data_ptr(), dy_pitch=_pitch(dy), dy_coff=0, dy_ps=int(dy_ps),
                    N=N, H=H, W=W, Cin=Cin, Cout=Cout, KH=k, KW=k, dwp=0, dbp=0, nslabs=0, dtype=_DT[x.dtype],
                    cout_real=int(cout) if (ps_r <= 1 and not x_ps and not dy_ps) else 0)
    nslabs = L.load().srk_wgrad_slabs(a)
    couts = L.load().srk_wgrad_slab_cout(a)     # channels per slab row: Cout, or 4 (compact slabs: large kernel, <= 4 real output channels)
    per = k * k * Cin * couts
    if nslabs > 0:      # slab mode: every workgroup writes its own slab, nothing to zero
        scratch = torch.empty(nslabs * (per + couts), dtype=torch.float32, device=dev)
    else:               # atomic mode: one zeroed slab
        scratch = torch.zeros(per + couts, dtype=torch.float32, device=dev)
    ns = max(nslabs, 1)
    dbp = scratch[ns * per:]
    a.dwp, a.dbp, a.nslabs = scratch.data_ptr(), (dbp.data_ptr() if want_bias else 0), nslabs
    L.call("srk_conv2d_wgrad", a, _stream())
    dw = out_w if out_w is not None else torch.empty(w_shape, dtype=torch.float32, device=dev)
    db = (out_b if out_b is not None else torch.empty(cout, dtype=torch.float32, device=dev)) if want_bias else None
    # the unfolded head conv presents its OIHW weight as a 1x1 conv over Cin*KH*KW channels
    f = L.WgradFinArgs(dwp=scratch.data_ptr(), dbp=dbp.data_ptr() if want_bias else 0, nslabs=nslabs, dw=dw.data_ptr(), db=_ptr(db),
                       Cout=cout, Cin=(cin * kh * kw) // (k * k), KH=k, KW=k, CinP=Cin, CoutP=couts,
                       ps_r=int(ps_r), scale=float(scale), accumulate=0)
    L.call("srk_wgrad_finalize", f, _stream())
    return dw, db


# --------------------------------------------------------------------------------------------
# deferred, grouped weight gradients
# --------------------------------------------------------------------------------------------
class _WgradQueue:
    """Weight-gradient jobs of the backward pass in flight (one process per GPU: autograd's device thread appends,
    the engine's final callback -- or a gradient-bucket hook -- flushes)."""

    def __init__(self):
        self.jobs = []          # dicts, see wgrad()
        self.rjobs = []         # row-sum jobs (channel-attention parameter gradients), see defer_rowsum()
        self.pwjobs = []        # finalize steps of pointwise-pair weight gradients (WDSR _Block_B), see pw_wgrad_raw()
        self.armed = False      # somebody will flush (final callback queued, or inside hold_wgrads)
        self.enabled = True
        self.targets = {}       # address of a dw buffer -> number of jobs queued on it (weight sharing -> rounds)
        self.gen = 0            # backward-pass generation (bumped by every flush)
        self.stream = None      # stream of the backward pass (the flush launches there, whatever thread runs it)


_WQ = _WgradQueue()
_WG_BLOCKS_PER_JOB = 32
import os as _os
if _knob("SRK_NO_DEFER_WGRAD") == "1":      # A/B knob (tools/): every weight gradient as its own launch, like round 1
    _WQ.enabled = False


class hold_wgrads:
    """Context manager for code that calls `wgrad` OUTSIDE an autograd backward pass (tests, micro-benchmarks): jobs are
    queued inside the block and flushed as one grouped launch when it exits."""

    def __enter__(self):
        self.prev, _WQ.armed = _WQ.armed, True
        return self

    def __exit__(self, *exc):
        _WQ.armed = self.prev
        if not self.prev and exc[0] is None:
            flush_wgrads()


class hold_wgrads_discard:
    """Like `hold_wgrads`, but the queued jobs are DROPPED on exit: a backward pass without its weight-gradient launches (bench.py times
    a trunk's data-gradient launches this way; the parameters' .grad then hold unfilled buffers -- never use them)."""

    def __enter__(self):
        self.prev, _WQ.armed = _WQ.armed, True
        return self

    def __exit__(self, *exc):
        discard_wgrads()
        _WQ.armed = self.prev


def set_defer_wgrad(enabled):
    """Deferral switch (default on).  Off = every weight gradient is its own launch inside backward, which is what
    code that reads gradients from inside backward needs (torch's DistributedDataParallel reducer)."""
    prev = _WQ.enabled
    _WQ.enabled = bool(enabled)
    return prev


def _grad_slot(p, shape):
    """How a deferred job delivers the gradient of leaf `p`:  ('new', None)  -> a fresh tensor handed to autograd;
    ('acc', tensor) -> accumulate into the existing fp32 gradient, autograd gets None;  None -> cannot defer."""
    if p is None:
        return ("new", None)
    if isinstance(p, torch.Tensor) and p.__dict__.get("_srk_wn_proxy", False):
        return ("new", None)            # effective weight of a weight-normed conv: its consumer (WeightNormGroup's backward) flushes first
    if not (isinstance(p, torch.Tensor) and p.is_leaf and p.requires_grad):
        return None
    if getattr(p, "_backward_hooks", None):
        return None                     # a tensor hook reads the gradient inside backward
    hooks = getattr(p, "_post_accumulate_grad_hooks", None)
    if hooks and not p.__dict__.get("_srk_flush_aware", False):
        return None
    g = p.grad
    if g is None:
        return ("new", None)
    if g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == tuple(shape) and g.is_cuda:
        return ("acc", g)
    return None


def _pass_id():
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def _pass_slot(p, shape, device):
    """Like `_grad_slot(p, shape)` for the [C]-sized parameters of BatchNorm / PReLU, which one module instance may see TWICE in one
    forward (the reference's ResBlock appends the SAME norm / act instance behind both convs, common.py:94-100: SRResNet): returns
    (tensor the finalize step adds into, or None;  tensor to hand to autograd, or None).  The first use of a pass hands a fresh
    zero-free tensor to autograd and remembers it under the running backward pass's id; the second use ADDS into that tensor in
    its own finalize step (autograd has not consumed it yet: AccumulateGrad runs after every use delivered) and hands over nothing --
    else autograd adds the two [C]-sized gradients with a launch of its own, 33 of them per SRResNet step."""
    sl = _grad_slot(p, shape)
    if sl is not None and sl[0] == "acc":
        return sl[1], None
    pid = _pass_id()
    if sl is None or pid < 0 or not isinstance(p, torch.Tensor):
        return None, "new"
    ent = p.__dict__.get("_srk_pass_grad")
    if ent is not None and ent[0] == pid:
        t = ent[1]()
        if t is not None and tuple(t.shape) == tuple(shape) and t.device == device:
            return t, None
    return None, "new+remember"


def _remember_pass_grad(p, t):
    p.__dict__["_srk_pass_grad"] = (_pass_id(), weakref.ref(t))


def _view_of(storage, shape, device, offset=0):
    return torch.empty(0, dtype=torch.float32, device=device).set_(storage, int(offset), tuple(shape))


def _arm_flush():
    """Make sure somebody flushes the queues: the autograd engine's final callback of the running backward pass."""
    if _WQ.armed:
        return True
    try:
        torch.autograd.Variable._execution_engine.queue_callback(flush_wgrads)
    except RuntimeError:            # not inside a backward pass (a Function's backward called by hand)
        return False
    _WQ.armed = True
    return True


class TableHolder:
    """Owns the device tables (job descriptors of the grouped launches) of hipGraphs captured under `static_tables`: keep it alive as
    long as the graphs, call `fence()` after the capture(s) and before the first replay."""

    ARENA_BYTES = 1 << 20

    def __init__(self):
        self.tables = []
        self.arena = None          # allocated by static_tables() OUTSIDE the capture: see take()
        self.used = 0

    def take(self, nbytes):
        """`nbytes` of the arena (16-byte aligned), or None when it is full.  NOT memory of the graph's own pool: an allocation made
        during the capture may reuse the address of an earlier temporary of the same graph, whose writer node would overwrite the
        table in every replay (the in-graph upload sits behind that writer; a table written once does not)."""
        n = (int(nbytes) + 15) // 16 * 16
        if self.arena is None or self.used + n > self.arena.numel():
            return None
        t = self.arena[self.used:self.used + n]
        self.used += n
        self.tables.append(t)
        return t

    def fence(self):
        if self.tables:
            L.check(L.load().srk_upload_fence(), "srk_upload_fence")


class _StaticTables:
    holder = None


@contextlib.contextmanager
def static_tables(holder):
    """Capture sites that own their graphs wrap the capture in this: the descriptor tables of the grouped launches (weight gradients,
    finalizes, row sums, weight normalisation) are then written ONCE, at capture time, instead of by upload launches inside the graph
    (3 per EDSR step, 30 per RCAN step at batch 16: include/srk.h, srk_upload_eager).  Valid because every address a replay sees is the
    capture's; the holder keeps the tables' memory from being reused inside the graph's pool.  Foreign captures (a user's own
    torch.cuda.graph around a step) keep the in-graph uploads."""
    ok = _STATIC_TABLES
    if ok:
        try:
            L.check(L.load().srk_upload_prepare(), "srk_upload_prepare")
        except RuntimeError:          # an older library loaded through SRK_LIB_PATH (A/B runs)
            ok = False
    if ok and holder.arena is None and not torch.cuda.is_current_stream_capturing():
        holder.arena = torch.empty(holder.ARENA_BYTES, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
        # the arena comes from the caching allocator on the ambient stream and is written from the library's upload stream: whatever that
        # memory was last used for must have finished first (explicit, not a side effect of torch.cuda.graph's own synchronize)
        torch.cuda.current_stream().synchronize()
    prev, _StaticTables.holder = _StaticTables.holder, (holder if ok else None)
    try:
        yield holder
    finally:
        _StaticTables.holder = prev


@contextlib.contextmanager
def graph_capture(g, **kw):
    """`torch.cuda.graph(g, **kw)` for a graph whose owner is this package: the grouped launches' tables are static (static_tables),
    kept alive by the graph object itself."""
    holder = TableHolder()
    try:
        with static_tables(holder):
            with torch.cuda.graph(g, **kw):
                yield holder
    finally:
        holder.fence()
    g._srk_tables = holder


def _upload_table(host_addr, nbytes, alloc, dev, st):
    """Host bytes -> a fresh device table of `alloc` bytes on stream `st` (see static_tables)."""
    h = _StaticTables.holder
    if h is not None and h.arena is not None and h.arena.device == torch.device(dev) and torch.cuda.is_current_stream_capturing():
        table = h.take(alloc)
        if table is not None:
            L.check(L.load().srk_upload_eager(table.data_ptr(), host_addr, nbytes), "srk_upload_eager")
            return table
    table = torch.empty(alloc, dtype=torch.uint8, device=dev)
    L.check(L.load().srk_upload_small(table.data_ptr(), host_addr, nbytes, st), "srk_upload_small")
    return table


def _launch_rowsums(rjobs, st):
    import ctypes as C
    n = len(rjobs)
    host = (L.RowsumJob * n)()
    for i, j in enumerate(rjobs):
        host[i].src, host[i].dst, host[i].n, host[i].k = j["src"].data_ptr(), j["dst"], j["n"], j["k"]
    nbytes = C.sizeof(L.RowsumJob) * n
    table = _upload_table(C.addressof(host), nbytes, _roundup(nbytes, 16), rjobs[0]["src"].device, st)
    L.check(L.load().srk_rowsum_group(table.data_ptr(), n, max(j["k"] for j in rjobs), st), "srk_rowsum_group")


def _launch_pw_finalize(pwjobs, st):
    """ONE launch sums the slabs of every queued pointwise-pair weight gradient (srk_pw_wgrad_finalize_group)."""
    import ctypes as C
    n = len(pwjobs)
    host = (L.PwWgradArgs * n)(*[j["a"] for j in pwjobs])
    nbytes = C.sizeof(L.PwWgradArgs) * n
    table = _upload_table(C.addressof(host), nbytes, _roundup(nbytes, 16), pwjobs[0]["keep"][0].device, st)
    items = max((j["a"].Chid * (j["a"].Cin + j["a"].CoutP)) // 4 + j["a"].Chid + j["a"].CoutP for j in pwjobs)
    bpj = max(1, min((items + 255) // 256, 256))           # one item per thread: the launch is a 30 MB-per-job read, latency-bound with fewer blocks
    L.check(L.load().srk_pw_wgrad_finalize_group(table.data_ptr(), n, bpj, st), "srk_pw_wgrad_finalize_group")
    for j in pwjobs:
        j["keep"].append(table)


def defer_rowsum(per, params, shapes_offsets):
    """Sum `per` [n][K] over n into a fresh [K] buffer whose slices become the gradients of `params` -- deferred to the end of
    the backward pass, where ONE launch serves every queued job (an RCAN backward has 200 of them).

    shapes_offsets[i] = (shape, offset into the K floats) of parameter i's gradient.  Returns the list of gradient tensors
    (views of the unfilled buffer, which autograd adopts as `.grad`), or None when deferral does not apply (a gradient
    already exists and autograd would read the unfilled buffer, hooks, a second use of the parameters in this pass...)."""
    if not _WQ.enabled or per.shape[0] == 0:
        return None
    for p, (shape, _) in zip(params, shapes_offsets):
        sl = _grad_slot(p, shape)
        if sl is None or sl[0] != "new" or p is None:
            return None
        seen = p.__dict__.get("_srk_pending_rs")
        if seen is not None and seen == _WQ.gen:
            # used twice in this pass: autograd will ADD this gradient to the (still unfilled) one queued earlier, right
            # after this backward returns -- fill the queued ones now (stream order puts the sums in front of that add)
            rj, _WQ.rjobs = _WQ.rjobs, []
            if rj:
                with torch.cuda.stream(_WQ.stream):
                    _launch_rowsums(rj, _WQ.stream.cuda_stream)
            return None
    if not _arm_flush():
        return None
    n, k = per.shape
    tot = torch.empty(k, dtype=torch.float32, device=per.device)
    stg = tot.untyped_storage()
    outs, new = [], []
    for p, (shape, off) in zip(params, shapes_offsets):
        numel = 1
        for d in shape:
            numel *= d
        v = tot[off:off + numel].view(shape)
        new.append((p, v.data_ptr(), stg, tuple(shape), off))
        p.__dict__["_srk_pending_rs"] = _WQ.gen
        outs.append(v)
    _WQ.rjobs.append(dict(src=per, dst=tot.data_ptr(), n=int(n), k=int(k), keep=[per, stg], new=new))
    _WQ.stream = torch.cuda.current_stream()
    del tot
    return outs


_ONES = {}


def backward(loss):
    """`loss.backward()` for a scalar loss, with the seed gradient taken from a cached tensor: the autograd engine otherwise fills a
    fresh `ones_like(loss)` per pass -- a 4.6 us launch of a 0.9 ms batch-16 step (it is a node of the captured step like any other)."""
    key = (loss.device, loss.dtype)
    one = _ONES.get(key)
    if one is None:
        if loss.is_cuda and torch.cuda.is_current_stream_capturing():
            # a tensor made inside a capture lives in that graph's pool and is only FILLED when the graph replays: not something to cache
            # for eager passes and other graphs -- this pass pays the fill, the next eager pass creates the cached one
            loss.backward()
            return
        one = _ONES[key] = torch.ones((), dtype=loss.dtype, device=loss.device)
    loss.backward(gradient=one if loss.dim() == 0 else one.expand_as(loss))


def discard_wgrads():
    """Drop whatever a backward pass that did NOT end normally left queued (an exception inside backward, a failed hipGraph
    capture: the engine's final callback may never have run, so `armed` would stay set and later passes would queue jobs
    nobody flushes).  Call before starting a fresh step."""
    _WQ.jobs, _WQ.rjobs, _WQ.pwjobs, _WQ.armed, _WQ.targets = [], [], [], False, {}
    _WQ.gen += 1


def flush_wgrads():
    """Launch every queued weight gradient: ONE grouped slab kernel per dtype and ONE grouped finalize per round."""
    import ctypes as C
    jobs, _WQ.jobs, _WQ.armed, _WQ.targets = _WQ.jobs, [], False, {}
    rjobs, _WQ.rjobs = _WQ.rjobs, []
    pwjobs, _WQ.pwjobs = _WQ.pwjobs, []
    _WQ.gen += 1
    if not jobs and not rjobs and not pwjobs:
        return
    lib = L.load()
    stream = _WQ.stream if _WQ.stream is not None else torch.cuda.current_stream()
    with torch.cuda.stream(stream):         # the flush may run on another thread than the backward nodes: same stream
        st = stream.cuda_stream
        if rjobs:
            _launch_rowsums(rjobs, st)
        if pwjobs:
            _launch_pw_finalize(pwjobs, st)
        for dt in sorted({j["a"].dtype for j in jobs}):
            grp = [j for j in jobs if j["a"].dtype == dt]
            n = len(grp)
            arr = (L.WgradArgs * n)(*[j["a"] for j in grp])
            nblocks, sfl = C.c_int(0), C.c_longlong(0)
            L.check(lib.srk_wgrad_group_plan(arr, n, None, None, None, C.byref(nblocks), C.byref(sfl)), "srk_wgrad_group_plan")
            dev = grp[0]["keep"][0].device
            scratch = torch.empty(sfl.value, dtype=torch.float32, device=dev)
            jb = lib.srk_wgrad_group_job_bytes()
            off_bj = _roundup(n * jb, 16)
            off_fin = _roundup(off_bj + 4 * nblocks.value, 16)
            fin_sz = C.sizeof(L.WgradFinArgs)
            total = _roundup(off_fin + n * fin_sz, 16)
            host = (C.c_ubyte * total)()
            base = C.addressof(host)
            L.check(lib.srk_wgrad_group_plan(arr, n, scratch.data_ptr(), base, base + off_bj, C.byref(nblocks), C.byref(sfl)),
                    "srk_wgrad_group_plan")
            # finalize table ordered by round (a job that accumulates into a buffer another job of this pass writes comes later)
            order = sorted(range(n), key=lambda i: grp[i]["round"])
            rounds = {}
            for pos, i in enumerate(order):
                a, f = arr[i], grp[i]
                fa = L.WgradFinArgs(dwp=a.dwp, dbp=a.dbp or 0, nslabs=a.nslabs, dw=f["dw"], db=f["db"], Cout=f["Cout"], Cin=f["Cin"],
                                    KH=3, KW=3, CinP=a.Cin, CoutP=a.Cout, ps_r=f["ps_r"], scale=f["scale"], accumulate=f["acc"])
                C.memmove(base + off_fin + pos * fin_sz, C.addressof(fa), fin_sz)
                rounds.setdefault(f["round"], [pos, 0])[1] += 1
            table = _upload_table(base, total, total, dev, st)
            L.check(lib.srk_conv2d_wgrad_group(table.data_ptr(), table.data_ptr() + off_bj, nblocks.value, dt, st), "srk_conv2d_wgrad_group")
            # workgroups per job: one per (2 input channels x 64 output channels) tile of the largest job, 32..256
            tiles = max(((a_.Cin + 1) // 2) * ((a_.Cout + 63) // 64) for a_ in arr)
            bpj = max(_WG_BLOCKS_PER_JOB, min(256, tiles))
            for r in sorted(rounds):
                pos, cnt = rounds[r]
                L.check(lib.srk_wgrad_finalize_group(table.data_ptr() + off_fin + pos * fin_sz, cnt, bpj, st), "srk_wgrad_finalize_group")
        # a gradient autograd COPIED instead of adopting (create_graph, layout contract) holds the bytes of the then
        # unfilled buffer: refresh it from the filled one.  ('new' jobs only: .grad was None, so the copy is all it holds)
        with torch.no_grad():
            for j in jobs + rjobs:
                for ent in j["new"]:
                    p, ptr, stg, shape = ent[:4]
                    if p is not None and p.__dict__.get("_srk_wn_proxy", False):
                        continue                    # non-leaf: the gradient went to WeightNormGroup's backward by address
                    g = p.grad if p is not None else None
                    if g is not None and g.data_ptr() != ptr and tuple(g.shape) == tuple(shape):
                        g.copy_(_view_of(stg, shape, g.device, ent[4] if len(ent) > 4 else 0))
    # `table`, `scratch`, operands and result storages are referenced by enqueued work only from here on: the caching
    # allocator re-issues a freed block on this stream behind these launches


def wgrad(x, dy, *, wparam=None, bparam=None, **kw):
    """Weight (+ bias) gradient of one conv.  3x3 16-bit convs whose parameters are leaves are QUEUED and computed by
    one grouped launch when the backward pass ends (`flush_wgrads`, the autograd engine's final callback); everything
    else runs now (`wgrad_raw`).  Returns what backward() hands to autograd for (weight, bias).

    A queued job returns EMPTY tensors that autograd adopts as `.grad` (AccumulateGrad takes over a gradient nobody else
    references); the queue keeps their storages -- not the tensors -- alive and the flush fills them by address."""
    want_bias = kw.get("want_bias", True)
    k, w_shape = kw["k"], kw["w_shape"]
    tw = _grad_target(wparam, w_shape, x.device)
    tb = _grad_target(bparam, (w_shape[0],), x.device) if want_bias else None
    if not (_WQ.enabled and k == 3 and x.dtype in (torch.bfloat16, torch.float16) and kw["N"] > 0 and kw.get("x_ps", 0) <= 1):
        return wgrad_raw(x, dy, out_w=tw, out_b=tb, **kw)
    sw = _grad_slot(wparam, w_shape)
    sb = _grad_slot(bparam, (w_shape[0],)) if want_bias else ("new", None)
    if sw is None or sb is None or wparam is None or (want_bias and sb[0] != sw[0]) or _batch_chunks(kw["N"], x, dy) > 1:
        return wgrad_raw(x, dy, out_w=tw, out_b=tb, **kw)
    a = L.WgradArgs(x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, x_ps=int(kw.get("x_ps", 0)),
                    dy=dy.data_ptr(), dy_pitch=_pitch(dy), dy_coff=0, dy_ps=int(kw.get("dy_ps", 0)),
                    N=kw["N"], H=kw["H"], W=kw["W"], Cin=kw["Cin"], Cout=kw["Cout"], KH=3, KW=3, dwp=0, dbp=1 if want_bias else 0,
                    nslabs=0, dtype=_DT[x.dtype])
    if not L.load().srk_wgrad_group_ok(a):
        return wgrad_raw(x, dy, out_w=tw, out_b=tb, **kw)
    cout, cin = w_shape[0], w_shape[1]
    dev = x.device
    ret_w = ret_b = None
    keep = [x, dy]
    new = []
    if sw[0] == "new":
        seen = wparam.__dict__.get("_srk_pending")          # second use of a shared weight in this pass -> accumulate round
        if seen is not None and seen[0] == _WQ.gen:
            dw_ptr, db_ptr, acc = seen[1], seen[2], 1
        else:
            ret_w = tw if tw is not None else torch.empty(w_shape, dtype=torch.float32, device=dev)
            ret_b = (tb if tb is not None else torch.empty(cout, dtype=torch.float32, device=dev)) if want_bias else None
            dw_ptr, db_ptr, acc = ret_w.data_ptr(), _ptr(ret_b), 0
            sw_stg = ret_w.untyped_storage()
            keep.append(sw_stg)
            new.append((wparam, dw_ptr, sw_stg, w_shape, ret_w.storage_offset()))
            if ret_b is not None:
                sb_stg = ret_b.untyped_storage()
                keep.append(sb_stg)
                new.append((bparam, db_ptr, sb_stg, (cout,), ret_b.storage_offset()))
            wparam.__dict__["_srk_pending"] = (_WQ.gen, dw_ptr, db_ptr)
    else:
        dw_ptr, db_ptr, acc = sw[1].data_ptr(), (sb[1].data_ptr() if want_bias else 0), 1
        keep += [sw[1], sb[1]]
    rnd = _WQ.targets.get(dw_ptr, 0)                        # jobs on one buffer finalize in successive rounds
    _WQ.targets[dw_ptr] = rnd + 1
    _WQ.jobs.append(dict(a=a, dw=dw_ptr, db=db_ptr, Cout=cout, Cin=cin, ps_r=int(kw.get("ps_r", 0)), scale=float(kw.get("scale", 1.0)),
                         acc=acc, round=rnd, keep=keep, new=new))
    _WQ.stream = torch.cuda.current_stream()
    if not _arm_flush():                # not inside a backward pass (a Function's backward called by hand)
        flush_wgrads()
    return ret_w, ret_b


def unfold_raw(x, sub, k, dtype):
    """NCHW fp32 -> NHWC im2col [N,H,W,pad16(C*k*k)] (- sub[c]), the head-conv input."""
    _need_gpu(x)
    n, c, h, w = x.shape
    xs = x.detach()
    if xs.dtype != torch.float32 or not xs.is_contiguous():
        xs = xs.float().contiguous()
    ks = pad16(c * k * k)
    dst = torch.empty((n, h, w, ks), dtype=dtype, device=x.device)
    a = L.UnfoldArgs(x=xs.data_ptr(), sub=_ptr(sub), dst=dst.data_ptr(), dst_pitch=ks, dst_coff=0,
                     N=n, Cin=c, H=h, W=w, KH=k, KW=k, Kstore=ks, dtype=_DT[dtype])
    L.call("srk_unfold_nchw", a, _stream())
    return dst


def to_nhwc(src, dtype, *, ps_r=0, scale=1.0):
    """NCHW fp32 -> NHWC dtype (channels padded to 16).  ps_r>1 un-shuffles:
    dst[n,y,x,c*r*r+i*r+j] = src[n,c,y*r+i,x*r+j]."""
    _need_gpu(src)
    r = ps_r if ps_r > 1 else 1
    n, cs, hs, ws = src.shape
    c, h, w = cs * r * r, hs // r, ws // r
    s = src.detach()
    if s.dtype != torch.float32 or not s.is_contiguous():
        s = s.float().contiguous()
    dst = torch.empty((n, h, w, pad16(c)), dtype=dtype, device=src.device)
    a = L.ToNhwcArgs(src=s.data_ptr(), dst=dst.data_ptr(), dst_pitch=dst.shape[3], dst_coff=0,
                     N=n, C=c, H=h, W=w, Cstore=dst.shape[3], ps_r=int(ps_r), scale=float(scale), dtype=_DT[dtype])
    L.call("srk_nchw_to_nhwc", a, _stream())
    return dst


def to_nchw(src, C=None):
    """NHWC dtype -> NCHW fp32 (first C channels)."""
    _need_gpu(src)
    n, h, w, cp = src.shape
    C = cp if C is None else C
    dst = torch.empty((n, C, h, w), dtype=torch.float32, device=src.device)
    a = L.ToNchwArgs(src=src.data_ptr(), src_pitch=_pitch(src), src_coff=0, dst=dst.data_ptr(),
                     N=n, C=C, H=h, W=w, dtype=_DT[src.dtype])
    L.call("srk_nhwc_to_nchw", a, _stream())
    return dst


class ToNhwcFn(torch.autograd.Function):
    """NCHW float -> NHWC `dtype`, channels padded to 16 (the inverse of ToNchwFn; backward is the other kernel)."""

    @staticmethod
    def forward(ctx, x, dtype):
        ctx.c = x.shape[1]
        return to_nhwc(x, dtype)

    @staticmethod
    def backward(ctx, g):
        return to_nchw(g.contiguous(), ctx.c), None


class ToNchwFn(torch.autograd.Function):
    """NHWC (padded channels) -> NCHW fp32, first C channels."""

    @staticmethod
    def forward(ctx, x, C):
        ctx.dt = x.dtype
        return to_nchw(x, C)

    @staticmethod
    def backward(ctx, g):
        return to_nhwc(g, ctx.dt), None


def nchw_to_nhwc(x, dtype):
    return ToNhwcFn.apply(x, dtype)


def nhwc_to_nchw(x, C):
    return ToNchwFn.apply(x, int(C))


def _f32c(t):
    t = t.detach()
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


# --------------------------------------------------------------------------------------------
# autograd Functions
# --------------------------------------------------------------------------------------------
_RES_LINK = _knob("SRK_NO_RES_LINK", "0") != "1"        # A/B knob
_STATIC_TABLES = _knob("SRK_NO_STATIC_TABLES", "0") != "1"      # A/B knob: descriptor tables written once per captured graph (static_tables)


class ResLink:
    """Ties the residual add at the END of a block to the FIRST conv of the block (SRResNet's ResBlock: x -> conv -> norm -> act -> conv ->
    norm, += x; common.py:74-109): in backward the block's input receives the upstream gradient g (through the add) plus the first
    conv's data gradient -- autograd would add the two with a launch of its own.  With a link the op that owns the residual add
    (`BatchNormFn` with `res`) parks g here instead of returning it, and the first conv's data-gradient launch adds it (`res`).  The park
    only happens when that conv's backward will run with its input gradient wanted (`armed`, decided in its forward)."""
    __slots__ = ("armed", "g")

    def __init__(self):
        self.armed, self.g = False, None


class ConvFn(torch.autograd.Function):
    """y = conv_same(x, w, b) * scale (+ res), optional fused PixelShuffle(ps_r) store.

    forward : DefaultConv2d (models/common.py:7-30) [+ nn.PixelShuffle, common.py:133] [+ `res += x`,
              edsr.py:46-47, rcan.py:72-73,121-122, rdn.py:109]
    backward: dgrad = the same kernel on flipped/transposed weights reading dy through the
              pixel-shuffle addressing; wgrad = srk_conv2d_wgrad."""

    @staticmethod
    def forward(ctx, x, w, b, res, scale, ps_r, link=None):
        _need_gpu(x)
        n, h, wd, cinp = x.shape
        cout, cin, k, _ = w.shape
        assert cinp == pad16(cin), f"input has {cinp} channels, conv expects pad16({cin})"
        ctx.link = link
        if link is not None:
            link.armed = bool(ctx.needs_input_grad[0]) and int(ps_r) <= 1
        pk = pack_conv(w, b, x.dtype, ps_r=ps_r)
        if ps_r > 1:
            c = cout // (ps_r * ps_r)
            assert c % 16 == 0, "fused PixelShuffle store needs C % 16 == 0"
            out = torch.empty((n, h * ps_r, wd * ps_r, c), dtype=x.dtype, device=x.device)
            conv_raw(x, pk, N=n, H=h, W=wd, Cin=cinp, Cout=cout, out=out, out_mode=L.OUT_NHWC_PS, ps_r=ps_r,
                     scale=scale, res=res)
        else:
            out = torch.empty((n, h, wd, pad16(cout)), dtype=x.dtype, device=x.device)
            conv_raw(x, pk, N=n, H=h, W=wd, Cin=cinp, Cout=out.shape[3], out=out, scale=scale, res=res)
        ctx.save_for_backward(x, w)
        ctx.cfg = (scale, ps_r, b is not None, res is not None)
        ctx.wb = (w, b)
        ctx.pg = _tok()
        ctx.gacc = x.__dict__.get("_srk_gacc")          # x is a prefix of a SliceBuffer whose consumers' gradients meet in one buffer
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        scale, ps_r, has_b, has_res = ctx.cfg
        g = g.contiguous()
        n, h, wd, cinp = x.shape
        cout, cin, k, _ = w.shape
        coutp = pad16(cout)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            pkd = pack_conv(w, None, x.dtype, dgrad=True, ps_r=ps_r, token=ctx.pg)
            if ctx.gacc is not None and ps_r <= 1:
                gx, add = ctx.gacc[0].grad_dest(ctx.gacc[1])       # the data gradient lands in (is added to) the shared gradient buffer
                conv_raw(g, pkd, N=n, H=h, W=wd, Cin=coutp, Cout=cinp, out=gx, scale=scale, use_bias=False, res=gx if add else None)
            else:
                gx = torch.empty_like(x)
                parked = None
                if ctx.link is not None and ctx.link.g is not None:        # the block's residual gradient: added by this launch
                    parked, ctx.link.g = ctx.link.g, None
                conv_raw(g, pkd, N=n, H=h, W=wd, Cin=coutp, Cout=cinp, out=gx, scale=scale, x_ps=ps_r, use_bias=False, res=parked)
        if ctx.needs_input_grad[1]:
            gw, gb = wgrad(x, g, wparam=ctx.wb[0], bparam=ctx.wb[1], N=n, H=h, W=wd, Cin=cinp, Cout=coutp, k=k,
                           w_shape=tuple(w.shape), ps_r=ps_r, scale=scale, dy_ps=ps_r, want_bias=has_b)
        return gx, gw, gb, (g if has_res else None), None, None, None


def conv(x, w, b, *, res=None, scale=1.0, ps_r=0, link=None):
    return ConvFn.apply(x, w, b, res, float(scale), int(ps_r), link)


class HeadConvFn(torch.autograd.Function):
    """NCHW fp32 image -> NHWC features: (x - sub) conv_same w + b, via boundary im2col + 1x1 MFMA conv.

    sub_mean + head conv: models/common.py:58-71 (sign=-1), edsr.py:41-44, rcan.py:116-119;
    SFENet1: rdn.py:100; WDSR head: wdsr.py:103-108.  The input image gets no gradient."""

    @staticmethod
    def forward(ctx, x, w, b, sub, dtype):
        _need_gpu(x)
        cout, cin, k, _ = w.shape
        n, c, h, wd = x.shape
        assert c == cin
        xu = unfold_raw(x, sub, k, dtype)
        pk = _pack_head(w, b, dtype)
        out = torch.empty((n, h, wd, pad16(cout)), dtype=dtype, device=x.device)
        conv_raw(xu, pk, N=n, H=h, W=wd, Cin=xu.shape[3], Cout=out.shape[3], out=out)
        ctx.save_for_backward(xu, w)
        ctx.has_b = b is not None
        return out

    @staticmethod
    def backward(ctx, g):
        xu, w = ctx.saved_tensors
        g = g.contiguous()
        n, h, wd, kp = xu.shape
        gw = gb = None
        if ctx.needs_input_grad[1]:
            gw, gb = wgrad_raw(xu, g, N=n, H=h, W=wd, Cin=kp, Cout=g.shape[3], k=1, w_shape=tuple(w.shape),
                               want_bias=ctx.has_b)
        return None, gw, gb, None, None


def _pack_head(w, b, dtype):
    """Pack an OIHW weight as the 1x1 conv over Cin*KH*KW unfolded channels."""
    return pack_conv(w, b, dtype, as_1x1=True)


def head_conv(x, w, b, sub, dtype):
    return HeadConvFn.apply(x, w, b, sub, dtype)


class TailConvFn(torch.autograd.Function):
    """NHWC features -> NCHW fp32 image: conv_same [+ PixelShuffle(r)] [+ res] + post_add.

    EDSR/RCAN tail conv + add_mean (edsr.py:49-52, common.py:58-71 sign=+1); RDN UPNet last conv
    (rdn.py:94); WDSR tail conv + PixelShuffle + `x += s` + mean (wdsr.py:110-115)."""

    @staticmethod
    def forward(ctx, x, w, b, res, post_add, ps_r):
        _need_gpu(x)
        n, h, wd, cinp = x.shape
        cout, cin, k, _ = w.shape
        r = ps_r if ps_r > 1 else 1
        pk = pack_conv(w, b, x.dtype)          # planar store keeps torch's channel order
        out = torch.empty((n, cout // (r * r), h * r, wd * r), dtype=torch.float32, device=x.device)
        resc = None if res is None else _f32c(res)
        conv_raw(x, pk, N=n, H=h, W=wd, Cin=cinp, Cout=cout, out=out, out_mode=L.OUT_PLANAR, ps_r=ps_r,
                 res=resc, post_add=post_add)
        ctx.save_for_backward(x, w)
        ctx.cfg = (ps_r, b is not None, res is not None)
        ctx.wb = (w, b)
        ctx.pg = _tok()
        return out

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        ps_r, has_b, has_res = ctx.cfg
        n, h, wd, cinp = x.shape
        cout, cin, k, _ = w.shape
        dy = to_nhwc(g, x.dtype, ps_r=ps_r)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            pkd = pack_conv(w, None, x.dtype, dgrad=True, token=ctx.pg)
            gx = torch.empty_like(x)
            conv_raw(dy, pkd, N=n, H=h, W=wd, Cin=dy.shape[3], Cout=cinp, out=gx, use_bias=False)
        if ctx.needs_input_grad[1]:
            gw, gb = wgrad(x, dy, wparam=ctx.wb[0], bparam=ctx.wb[1], N=n, H=h, W=wd, Cin=cinp, Cout=dy.shape[3], k=k,
                           w_shape=tuple(w.shape), want_bias=has_b)
        return gx, gw, gb, (g if has_res else None), None, None


def tail_conv(x, w, b, *, res=None, post_add=None, ps_r=0):
    return TailConvFn.apply(x, w, b, res, post_add, int(ps_r))


class SkipConvFn(torch.autograd.Function):
    """NCHW fp32 image -> NCHW fp32 image: (x - sub) conv_same(k x k) + PixelShuffle(r), WDSR's skip
    branch (models/wdsr.py:90-94,107): boundary im2col + 1x1 MFMA conv with the planar shuffle store."""

    @staticmethod
    def forward(ctx, x, w, b, sub, ps_r, dtype):
        _need_gpu(x)
        cout, cin, k, _ = w.shape
        n, c, h, wd = x.shape
        r = ps_r if ps_r > 1 else 1
        xu = unfold_raw(x, sub, k, dtype)
        pk = _pack_head(w, b, dtype)
        out = torch.empty((n, cout // (r * r), h * r, wd * r), dtype=torch.float32, device=x.device)
        conv_raw(xu, pk, N=n, H=h, W=wd, Cin=xu.shape[3], Cout=cout, out=out, out_mode=L.OUT_PLANAR, ps_r=ps_r)
        ctx.save_for_backward(xu, w)
        ctx.cfg = (ps_r, b is not None)
        return out

    @staticmethod
    def backward(ctx, g):
        xu, w = ctx.saved_tensors
        ps_r, has_b = ctx.cfg
        n, h, wd, kp = xu.shape
        gw = gb = None
        if ctx.needs_input_grad[1]:
            dy = to_nhwc(g, xu.dtype, ps_r=ps_r)
            gw, gb = wgrad_raw(xu, dy, N=n, H=h, W=wd, Cin=kp, Cout=dy.shape[3], k=1, w_shape=tuple(w.shape),
                               want_bias=has_b)
        return None, gw, gb, None, None, None


def skip_conv(x, w, b, sub, ps_r, dtype):
    return SkipConvFn.apply(x, w, b, sub, int(ps_r), dtype)


# --------------------------------------------------------------------------------------------
# the last upsampling stage + tail conv as one 5x5 convolution (csrc/hr_tail.hip)
# --------------------------------------------------------------------------------------------
_HR_COLLAPSE = _knob("SRK_NO_HR_COLLAPSE", "0") != "1"       # A/B knob: 1 = the two layers one after the other


def hr_tail_ok(x, wu, wt, ps_r):
    """Whether `conv3x3(x; wu) -> PixelShuffle(2) -> conv3x3(. ; wt) -> NCHW fp32` runs as ONE collapsed 5x5 convolution: 16-bit
    storage, 64 input channels (what the direct large-kernel kernels take), <= 4 image channels, both kernels 3x3."""
    if not _HR_COLLAPSE or int(ps_r) != 2 or x.dtype not in (torch.bfloat16, torch.float16):
        return False
    cu, ci, ku, _ = wu.shape
    o, c, kt, _ = wt.shape
    if ku != 3 or kt != 3 or cu != 4 * c or ci != 64 or x.shape[3] != 64 or not 1 <= o <= 4:
        return False
    n, h, w, _ = x.shape
    return n > 0 and max(h, w) <= 512 and x.numel() * 2 < _ADDR_LIMIT and n * o * 4 * h * w < (1 << 31)


def _hr_bufs(o, ci, dev):
    f32 = torch.float32
    return dict(weff=torch.empty((4 * o, ci, 5, 5), dtype=f32, device=dev), beff=torch.empty(4 * o, dtype=f32, device=dev),
                wedge=torch.empty((4, 2 * o, ci, 5), dtype=f32, device=dev), bedge=torch.empty((4, 2 * o), dtype=f32, device=dev),
                wcor=torch.empty((4, o, ci), dtype=f32, device=dev), bcor=torch.empty((4, o), dtype=f32, device=dev))


class HrTailFn(torch.autograd.Function):
    """NHWC features -> NCHW fp32 image: conv3x3(Ci -> 4C) -> PixelShuffle(2) -> conv3x3(C -> O) [+ post_add] as ONE linear map.

    UpscaleBlock's last stage + the tail conv of EDSR / RCAN / RDN (models/common.py:112-139, edsr.py:48-52, rcan.py:102-104,
    rdn.py:85-95): no activation sits between the two convolutions, so the image is a 5x5 convolution Ci -> 4 O of x whose
    weights are sums of products of the two layers' weights (include/srk.h, csrc/hr_tail.hip) -- 8x fewer multiply-adds and no
    C-channel tensor at the doubled resolution, forward or backward.  The parameters of record stay the two layers' own
    (state_dict unchanged); their gradients come from the chain rule through the collapse, exact up to rounding, border pixels
    included (the tail conv's zero padding of the shuffled tensor is restored by `srk_hrtail_edge_*`)."""

    @staticmethod
    def _args(x, wu, bu, wt, bt, bufs, **kw):
        n, h, w, _ = x.shape
        o, c = wt.shape[0], wt.shape[1]
        return L.HrTailArgs(wt=wt.data_ptr(), bt=_ptr(bt), wu=wu.data_ptr(), bu=_ptr(bu), O=o, C=c, Ci=wu.shape[1],
                            weff=bufs["weff"].data_ptr(), beff=bufs["beff"].data_ptr(), wedge=bufs["wedge"].data_ptr(),
                            bedge=bufs["bedge"].data_ptr(), wcor=bufs["wcor"].data_ptr(), bcor=bufs["bcor"].data_ptr(),
                            x=x.data_ptr(), x_pitch=_pitch(x), N=n, H=h, W=w, dtype=_DT[x.dtype], **kw)

    @staticmethod
    def forward(ctx, x, wu, bu, wt, bt, post_add):
        _need_gpu(x)
        n, h, w, ci = x.shape
        o = wt.shape[0]
        dev, f32 = x.device, torch.float32
        wu_, wt_ = _f32c(wu), _f32c(wt)
        bu_, bt_ = (None if bu is None else _f32c(bu)), (None if bt is None else _f32c(bt))
        bufs = _hr_bufs(o, ci, dev)
        L.call("srk_hrtail_collapse", HrTailFn._args(x, wu_, bu_, wt_, bt_, bufs), _stream())
        pk = pack_conv(bufs["weff"], bufs["beff"], x.dtype, cache=False)
        out = torch.empty((n, o, 2 * h, 2 * w), dtype=f32, device=dev)
        conv_raw(x, pk, N=n, H=h, W=w, Cin=ci, Cout=4 * o, out=out, out_mode=L.OUT_PLANAR, ps_r=2, post_add=post_add)
        L.call("srk_hrtail_edge_fwd", HrTailFn._args(x, wu_, bu_, wt_, bt_, bufs, out=out.data_ptr()), _stream())
        ctx.save_for_backward(x, wu_, wt_, *( [bu_] if bu_ is not None else []))
        ctx.has = (bu is not None, bt is not None)
        ctx.bufs = bufs
        ctx.wb = (wu, bu, wt, bt)
        return out

    @staticmethod
    def backward(ctx, g):
        x, wu_, wt_ = ctx.saved_tensors[:3]
        has_bu, has_bt = ctx.has
        bu_ = ctx.saved_tensors[3] if has_bu else None
        bufs = ctx.bufs
        n, h, w, ci = x.shape
        o, c = wt_.shape[0], wt_.shape[1]
        dev, f32 = x.device, torch.float32
        g = _f32c(g)
        st = _stream()
        g12 = to_nhwc(g, x.dtype, ps_r=2)                      # [n, h, w, 16]: channel o*4 + a*2 + b = the un-shuffled gradient
        need_x = ctx.needs_input_grad[0]
        need_w = any(ctx.needs_input_grad[1:5])
        gx = None
        if need_x:
            pkd = pack_conv(bufs["weff"], None, x.dtype, dgrad=True, cache=False)
            gx = torch.empty_like(x)
            conv_raw(g12, pkd, N=n, H=h, W=w, Cin=g12.shape[3], Cout=ci, out=gx, use_bias=False)
            L.call("srk_hrtail_edge_bwd_x", HrTailFn._args(x, wu_, bu_, wt_, None, bufs, g=g.data_ptr(), dx=gx.data_ptr(), dx_pitch=_pitch(gx)), st)
        gwu = gbu = gwt = gbt = None
        if need_w:
            r, r0 = wgrad_raw(x, g12, N=n, H=h, W=w, Cin=ci, Cout=g12.shape[3], k=5, w_shape=(4 * o, ci, 5, 5), want_bias=True)
            red = dict(eedge=torch.empty((4, 2 * o, ci, 5), dtype=f32, device=dev), e0=torch.empty((4, 2 * o), dtype=f32, device=dev),
                       ecor=torch.empty((4, o, ci), dtype=f32, device=dev), k0=torch.empty((4, o), dtype=f32, device=dev))
            scratch = torch.empty(int(L.load().srk_hrtail_scratch_floats(n, ci)), dtype=f32, device=dev)
            ptrs = {k: v.data_ptr() for k, v in red.items()}
            L.call("srk_hrtail_edge_bwd_w", HrTailFn._args(x, wu_, bu_, wt_, None, bufs, g=g.data_ptr(), scratch=scratch.data_ptr(), **ptrs), st)
            # (written straight into trainer.GradSync's flat buffer when the parameters name a slice of it: _grad_target)
            def dest(p, shape):
                t = _grad_target(p, shape, dev) if p is not None and p.dtype == f32 else None
                return t if t is not None else torch.empty(shape, dtype=f32, device=dev)
            pu, pbu, pt, pbt = ctx.wb
            gwu, gwt = dest(pu, tuple(wu_.shape)), dest(pt, tuple(wt_.shape))
            gbu = dest(pbu, (4 * c,)) if has_bu else None
            gbt = dest(pbt, (o,)) if has_bt else None
            L.call("srk_hrtail_expand", HrTailFn._args(x, wu_, bu_, wt_, None, bufs, r=r.data_ptr(), r0=r0.data_ptr(), dwt=gwt.data_ptr(),
                                                       dbt=_ptr(gbt), dwu=gwu.data_ptr(), dbu=_ptr(gbu), **ptrs), st)
        return gx, gwu, gbu, gwt, gbt, None


def hr_tail(x, wu, bu, wt, bt, *, post_add=None):
    return HrTailFn.apply(x, wu, bu, wt, bt, post_add)


class ConvChainFn(torch.autograd.Function):
    """out = chain(x) * scale + x, chain = conv_1 [ReLU] conv_2 [ReLU] ... conv_L  (residual blocks).

    ResBlock (models/common.py:74-109), WDSR _Block_A (wdsr.py:9-27) and _Block_B (wdsr.py:30-51).
    `relus[i]` says whether a ReLU follows conv i (never after the last).  Saved for backward: x and
    the post-ReLU activations; the ReLU backward is the `mask` epilogue of the dgrad that produces
    the gradient of that activation, `* scale` and `+ g` are dgrad epilogues too."""

    @staticmethod
    def forward(ctx, x, scale, relus, *params):
        _need_gpu(x)
        L_ = len(relus)
        ws, bs = params[0::2], params[1::2]
        n, h, wd, _ = x.shape
        acts = [x]
        bits = []
        a = x
        paired = L_ == 2 and tuple(relus) == (True, False) and pair_ok(x, ws[0], ws[1])
        if paired:      # small batch: the whole block is one launch, the intermediate goes to HBM only for the backward pass
            y1 = torch.empty_like(x)
            a = conv_pair_raw(x, pack_conv(ws[0], bs[0], x.dtype), pack_conv(ws[1], bs[1], x.dtype), out=torch.empty_like(x),
                              relu_mid=True, mid=y1, scale_out=scale, res=x)
            acts.append(y1)
        for i in range(0 if paired else L_):
            w = ws[i]
            cout, cin, k, _ = w.shape
            pk = pack_conv(w, bs[i], x.dtype)
            out = torch.empty((n, h, wd, pad16(cout)), dtype=x.dtype, device=x.device)
            last = i == L_ - 1
            conv_raw(a, pk, N=n, H=h, W=wd, Cin=a.shape[3], Cout=out.shape[3], out=out, relu=relus[i],
                     scale=scale if last else 1.0, res=x if last else None,
                     relu_bits="want" if (relus[i] and not last and ctx.needs_input_grad[0]) else None)
            a = out
            if not last:
                acts.append(a)
                bits.append(out.__dict__.pop("_srk_bits", None))
        ctx.bits = bits                           # bits[i]: ReLU sign bits of acts[i + 1] (4 bytes per pixel and half instead of 64), or None
        ctx.save_for_backward(*acts, *ws)
        ctx.cfg = (scale, tuple(relus), tuple(b is not None for b in bs))
        ctx.wb = (tuple(ws), tuple(bs))
        ctx.pg = _tok()
        return a

    @staticmethod
    def backward(ctx, g):
        scale, relus, has_b = ctx.cfg
        L_ = len(relus)
        acts, ws = ctx.saved_tensors[:L_], ctx.saved_tensors[L_:]
        g = g.contiguous()
        n, h, wd, _ = g.shape
        grads = [None] * (2 * L_)
        dy = g
        paired = L_ == 2 and relus == (True, False) and pair_ok(g, ws[0], ws[1])
        if paired:      # dgrad 2 (ReLU mask, * scale) and dgrad 1 (+ g) in one launch; the two weight gradients as usual
            x, y1 = acts
            g1 = torch.empty_like(g)
            gx = conv_pair_raw(g, pack_conv(ws[1], None, g.dtype, dgrad=True, token=ctx.pg),
                               pack_conv(ws[0], None, g.dtype, dgrad=True, token=ctx.pg), out=torch.empty_like(g),
                               scale_mid=scale, mask=y1, mid=g1, res=g, use_bias=False)
            for i, (a_in, d, sc) in enumerate(((x, g1, 1.0), (y1, g, scale))):
                if ctx.needs_input_grad[3 + 2 * i]:
                    grads[2 * i], grads[2 * i + 1] = wgrad(
                        a_in, d, wparam=ctx.wb[0][i], bparam=ctx.wb[1][i], N=n, H=h, W=wd, Cin=64, Cout=64, k=3,
                        w_shape=tuple(ws[i].shape), scale=sc, want_bias=has_b[i])
            return (gx, None, None, *grads)
        for i in range(L_ - 1, -1, -1):
            w = ws[i]
            a_in = acts[i]
            k = w.shape[2]
            sc = scale if i == L_ - 1 else 1.0
            if ctx.needs_input_grad[3 + 2 * i]:
                gw, gb = wgrad(a_in, dy, wparam=ctx.wb[0][i], bparam=ctx.wb[1][i], N=n, H=h, W=wd, Cin=a_in.shape[3],
                               Cout=dy.shape[3], k=k, w_shape=tuple(w.shape), scale=sc, want_bias=has_b[i])
                grads[2 * i], grads[2 * i + 1] = gw, gb
            pkd = pack_conv(w, None, g.dtype, dgrad=True, token=ctx.pg)
            gin = torch.empty_like(a_in)
            mk = a_in if (i > 0 and relus[i - 1]) else None
            mb = ctx.bits[i - 1] if (mk is not None and i - 1 < len(ctx.bits)) else None
            conv_raw(dy, pkd, N=n, H=h, W=wd, Cin=dy.shape[3], Cout=a_in.shape[3], out=gin, scale=sc,
                     res=g if i == 0 else None, mask=mk, mask_bits=mb, use_bias=False)
            dy = gin
        return (dy, None, None, *grads)


def conv_chain(x, convs, relus, scale=1.0):
    """convs: list of (weight, bias) ; relus: list of bool (ReLU after conv i)."""
    flat = []
    for w, b in convs:
        flat += [w, b]
    return ConvChainFn.apply(x, float(scale), tuple(bool(r) for r in relus), *flat)


# --------------------------------------------------------------------------------------------
# a whole residual trunk per launch (csrc/conv_igemm.hip: conv_trunk_kernel; include/srk.h: srk_conv_trunk)
# --------------------------------------------------------------------------------------------
_TRUNK_OFF = _knob("SRK_NO_TRUNK", "0") == "1"       # A/B knob: the trunk as one srk_conv2d launch per convolution


def _trunk_layer(x, pk, out, *, relu=False, scale=1.0, res=None, relu_bits=None, mask_bits=None, use_bias=True):
    n, h, w, _ = x.shape
    return L.ConvArgs(x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, x_ps=0, N=n, H=h, W=w, Cin=64, wpk=pk.wpk.data_ptr(),
                      bias=_ptr(pk.bias) if use_bias else 0, CoutP=pk.CoutP, Cout=64, KH=3, KW=3, relu=int(relu), scale=float(scale), res=_ptr(res),
                      res_pitch=0 if res is None else _pitch(res), res_coff=0, mask=0, mask_pitch=0, mask_coff=0, mask_from=0, out=out.data_ptr(),
                      out_pitch=_pitch(out), out_coff=0, out_mode=L.OUT_NHWC, ps_r=0, post_add=0, dtype=_DT[x.dtype], cout_real=0,
                      relu_bits=_ptr(relu_bits), mask_bits=_ptr(mask_bits))


def _trunk_add_layer(x, res, out):
    n, h, w, _ = x.shape
    return L.ConvArgs(x=x.data_ptr(), x_pitch=64, x_coff=0, x_ps=0, N=n, H=h, W=w, Cin=64, wpk=0, bias=0, CoutP=64, Cout=64, KH=0, KW=0, relu=0, scale=1.0,
                      res=res.data_ptr(), res_pitch=64, res_coff=0, mask=0, mask_pitch=0, mask_coff=0, mask_from=0, out=out.data_ptr(), out_pitch=64,
                      out_coff=0, out_mode=L.OUT_NHWC, ps_r=0, post_add=0, dtype=_DT[x.dtype], cout_real=0, relu_bits=0, mask_bits=0)


def _trunk_launch(layers, dev):
    """The table of `layers` (ConvArgs) to the device, one srk_conv_trunk launch.  False (nothing launched) when the table does not qualify."""
    import ctypes as C
    nl = len(layers)
    host = (L.ConvArgs * nl)(*layers)
    lib = L.load()
    if not lib.srk_conv_trunk_ok(host, nl):
        return False
    st = _stream()
    nbytes = C.sizeof(L.ConvArgs) * nl
    table = _upload_table(C.addressof(host), nbytes, _roundup(nbytes, 16), dev, st)
    L.check(lib.srk_conv_trunk(host, table.data_ptr(), nl, st), "srk_conv_trunk")
    return True


def res_trunk_ok(x, blocks, tail):
    """EDSR's body -- ResBlocks of two 3x3 F -> F convs with a ReLU between, then one more conv and the long skip -- as ONE launch: 16-bit
    NHWC features with F = 64, a batch that is a whole number of rounds over the CUs, not while a segmented backward is being recorded
    (its cuts lie between the blocks)."""
    if _TRUNK_OFF or not _SIGN_BITS or getattr(_TLS, "seg", None) is not None or not x.is_cuda or x.dtype not in (torch.bfloat16, torch.float16):
        return False
    if x.dim() != 4 or x.shape[3] != 64 or not x.is_contiguous() or not blocks:
        return False
    cus = L.load().srk_device_cus()
    if cus <= 0 or x.shape[0] < cus or x.shape[0] % cus != 0 or _batch_chunks(x.shape[0], x) != 1:
        return False
    ws = [w for blk in blocks for (w, _b) in blk] + [tail[0]]
    return all(len(blk) == 2 for blk in blocks) and all(tuple(w.shape) == (64, 64, 3, 3) for w in ws)


class ResTrunkFn(torch.autograd.Function):
    """r = conv_f(B_n(... B_1(x))) + x with B_i(t) = t + scale * conv_2(relu(conv_1(t))): EDSR's body (models/edsr.py:24-31,44-47; ResBlock:
    models/common.py:74-109).  Forward: ONE srk_conv_trunk launch (2 n + 1 layers per image, back to back on the CU that owns the image);
    backward: ONE launch for the 2 n + 1 data gradients and the long skip's add, the weight gradients queued as usual.  Saved: the block
    inputs, the post-ReLU activations and their sign bits -- what the per-block ConvChainFn saves.  Bit-identical to the per-layer launches."""

    @staticmethod
    def forward(ctx, x, scale, nb, *params):
        _need_gpu(x)
        dt, dev = x.dtype, x.device
        n, h, wd, _ = x.shape
        ws, bs = params[0::2], params[1::2]
        train = any(ctx.needs_input_grad)
        xs, hs, bits, layers = [x], [], [], []
        spare = []                                   # no gradient wanted: three buffers rotate (no layer writes a buffer it reads)

        def take():
            return spare.pop() if spare else torch.empty_like(x)
        cur = x
        for b in range(nb):
            hb, xo = take(), take()
            bt = torch.empty((n * h * wd, 2), dtype=torch.int32, device=dev) if train else None
            layers.append(_trunk_layer(cur, pack_conv(ws[2 * b], bs[2 * b], dt), hb, relu=True, relu_bits=bt))
            layers.append(_trunk_layer(hb, pack_conv(ws[2 * b + 1], bs[2 * b + 1], dt), xo, scale=scale, res=cur))
            if train:
                xs.append(xo)
                hs.append(hb)
                bits.append(bt)
            else:
                spare.append(hb)
                if cur is not x:
                    spare.append(cur)
            cur = xo
        out = take()
        layers.append(_trunk_layer(cur, pack_conv(ws[2 * nb], bs[2 * nb], dt), out, res=x))
        if not _trunk_launch(layers, dev):
            raise RuntimeError("ResTrunkFn: srk_conv_trunk_ok refused a table res_trunk_ok had accepted")
        if train:
            ctx.save_for_backward(*xs, *hs, *ws)
            ctx.bits = bits
            ctx.cfg = (float(scale), int(nb), tuple(b is not None for b in bs))
            ctx.wb = (tuple(ws), tuple(bs))
            ctx.pg = _tok()
        return out

    @staticmethod
    def backward(ctx, g):
        scale, nb, has_b = ctx.cfg
        sv = ctx.saved_tensors
        xs, hs, ws = sv[:nb + 1], sv[nb + 1:2 * nb + 1], sv[2 * nb + 1:]
        g = g.contiguous()
        dt, dev = g.dtype, g.device
        n, h, wd, _ = g.shape
        need_x = bool(ctx.needs_input_grad[0])

        def pkd(w):
            return pack_conv(w, None, dt, dgrad=True, token=ctx.pg)
        gxs = [None] * (nb + 1)                        # gxs[b]: gradient of block b's input (gxs[nb]: of the last conv's input)
        ghs = [None] * nb
        gxs[nb] = torch.empty_like(g)
        layers = [_trunk_layer(g, pkd(ws[2 * nb]), gxs[nb], use_bias=False)]
        for b in range(nb - 1, -1, -1):
            ghs[b] = torch.empty_like(g)
            layers.append(_trunk_layer(gxs[b + 1], pkd(ws[2 * b + 1]), ghs[b], scale=scale, mask_bits=ctx.bits[b], use_bias=False))
            if b > 0 or need_x:
                gxs[b] = torch.empty_like(g)
                layers.append(_trunk_layer(ghs[b], pkd(ws[2 * b]), gxs[b], res=gxs[b + 1], use_bias=False))
        gx = None
        if need_x:
            gx = torch.empty_like(g)
            layers.append(_trunk_add_layer(gxs[0], g, gx))          # the long skip: d r / d x = (chain) + identity
        if not _trunk_launch(layers, dev):
            raise RuntimeError("ResTrunkFn.backward: srk_conv_trunk_ok refused the data-gradient table")
        grads = [None] * (2 * (2 * nb + 1))
        jobs = [(2 * nb, xs[nb], g, 1.0)]
        for b in range(nb - 1, -1, -1):
            jobs += [(2 * b + 1, hs[b], gxs[b + 1], scale), (2 * b, xs[b], ghs[b], 1.0)]
        for i, a_in, dy, sc in jobs:
            if ctx.needs_input_grad[3 + 2 * i]:
                grads[2 * i], grads[2 * i + 1] = wgrad(a_in, dy, wparam=ctx.wb[0][i], bparam=ctx.wb[1][i], N=n, H=h, W=wd, Cin=64, Cout=64, k=3,
                                                       w_shape=(64, 64, 3, 3), scale=sc, want_bias=has_b[i])
        return (gx, None, None, *grads)


def res_trunk(x, blocks, tail, scale=1.0):
    """blocks: [((w1, b1), (w2, b2)), ...]; tail: (w, b).  Call only when res_trunk_ok(x, blocks, tail)."""
    flat = []
    for blk in blocks:
        for w, b in blk:
            flat += [w, b]
    flat += [tail[0], tail[1]]
    return ResTrunkFn.apply(x, float(scale), len(blocks), *flat)


# --------------------------------------------------------------------------------------------
# backward in SEGMENTS (multi-GPU: a bucket's all-reduce can start while the rest of backward still runs, also when the
# step is replayed from hipGraphs -- a capture cannot be switched from inside `backward()`, so the cut has to be in the graph)
# --------------------------------------------------------------------------------------------
class record_segments:
    """`with record_segments(every=k) as rec: loss = forward(...)`: while active, every k-th `ops.cut(...)` the model's forward
    passes DETACHES its tensors (the rest of the forward continues on leaf copies that share the memory), which splits the
    autograd graph into independent pieces; `rec.cuts` lists, in forward order, the (original, leaf) pairs of every kept cut.
    `backward_segments(loss, rec.cuts, after)` then runs the backward pass piece by piece, last layers first."""

    def __init__(self, every=1, limit=None):
        self.every, self.count, self.cuts, self.limit, self.taken, self.keeps = max(1, int(every)), 0, [], limit, 0, 0

    def __enter__(self):
        self.prev = getattr(_TLS, "seg", None)
        _TLS.seg = self
        return self

    def __exit__(self, *exc):
        _TLS.seg = self.prev


def cut(*ts, keep=False):
    """Segment boundary in a model's forward: identity, unless a `record_segments` block is active (then every k-th call
    detaches).  `keep=True` marks tensors that are consumed again FAR downstream (the head's output that is added back after
    the body, RDN's block outputs that all feed the global fusion, WDSR's skip branch): they are always detached while
    recording -- a long skip that is not a leaf would drag its producer into the top segment's pass and a second time into
    its own -- and their leaves collect the gradient contributions of every later consumer."""
    rec = getattr(_TLS, "seg", None)
    out = ts
    if rec is not None:
        if not keep:
            rec.count += 1
        else:
            rec.keeps += 1
        take = keep or (rec.count % rec.every == 0 and (rec.limit is None or rec.taken < rec.limit))
        if take and any(t.requires_grad for t in ts):
            if not keep:
                rec.taken += 1
            out = tuple(t.detach().requires_grad_(True) if t.requires_grad else t for t in ts)
            rec.cuts.append([(o, l) for o, l in zip(ts, out) if l is not o])
    return out[0] if len(out) == 1 else out


def backward_segments(loss, cuts, after=None):
    """`loss.backward()` as len(cuts) + 1 autograd passes: the part above the last cut first, then downwards cut by cut, each
    pass started from the cut's original tensors with the gradients its leaves have collected.  `after(k)` is called behind
    pass k (k = 0: the top of the network): the gradients of that piece's parameters are complete there."""
    loss.backward()
    if after is not None:
        after(0)
    for k, pairs in enumerate(reversed(cuts), 1):
        roots = [o for o, l in pairs if l.grad is not None]
        grads = [l.grad for o, l in pairs if l.grad is not None]
        if roots:
            torch.autograd.backward(roots, grads)
        if after is not None:
            after(k)


# --------------------------------------------------------------------------------------------
# weight normalisation of all weight-normed convs of a model in one launch per direction (csrc/wn.hip)
# --------------------------------------------------------------------------------------------
class _WnFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, grp, *params):
        grp._launch(params, backward=False)
        ctx.grp = grp
        ctx.save_for_backward(*params)
        return tuple(grp.flat[o:o + n].view(shape) for (o, n, shape, _, _) in grp.layout)     # fresh view objects every pass

    @staticmethod
    def backward(ctx, *grads):
        grp, params = ctx.grp, ctx.saved_tensors
        flush_wgrads()          # the deferred weight gradients of this pass are this node's inputs: compute them first
        dev = params[0].device
        dflat = torch.empty(grp.total, dtype=torch.float32, device=dev)
        dg = torch.empty(grp.rows, dtype=torch.float32, device=dev)
        gs = []
        for gr, (off, n, shape, r0, rows) in zip(grads, grp.layout):
            if gr is None:
                gr = torch.zeros(shape, dtype=torch.float32, device=dev)
            elif gr.dtype != torch.float32 or not gr.is_contiguous():
                gr = gr.float().contiguous()
            gs.append(gr)
        grp._launch(params, backward=True, dws=gs, dv=dflat, dg=dg)
        out = [None]
        for (off, n, shape, r0, rows) in grp.layout:
            out += [dflat[off:off + n].view(shape), dg[r0:r0 + rows].view(rows, *([1] * (len(shape) - 1)))]
        return tuple(out)


class WeightNormGroup:
    """The effective weights w = g * v / ||v|| (nn.utils.weight_norm, models/wdsr.py:62) of a list of weight-normed convs:
    ONE launch computes all of them into one flat buffer whose address never changes (so the grouped pack launch and a captured
    hipGraph can name the weights), ONE launch turns their gradients into those of `weight_v` / `weight_g`.  `weights()`
    returns one tensor per conv; each carries the identity of its `weight_v` for the packed-weight group and is accepted by
    the deferred (grouped) weight gradients although it is not a leaf."""

    def __init__(self, convs):
        self.convs = list(convs)
        self.flat = self.inv = None
        self.layout = []
        self.total = self.rows = 0
        self.key = None

    def _build(self, params):
        dev = params[0].device
        off = r0 = 0
        self.layout = []
        for v in params[0::2]:
            n, rows = v.numel(), v.shape[0]
            self.layout.append((off, n, tuple(v.shape), r0, rows))
            off += _roundup(n, 4)
            r0 += rows
        self.total, self.rows = off, r0
        self.flat = torch.empty(off, dtype=torch.float32, device=dev)
        self.inv = torch.empty(r0, dtype=torch.float32, device=dev)

    def _launch(self, params, backward, dws=None, dv=None, dg=None):
        import ctypes as C
        key = tuple(p.data_ptr() for p in params)
        if self.flat is None or self.key != key or self.flat.device != params[0].device:
            for p in params:
                if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                    raise RuntimeError("WeightNormGroup needs contiguous fp32 weight_v / weight_g on the GPU")
            if self.flat is None or self.flat.device != params[0].device or [tuple(v.shape) for v in params[0::2]] != [l[2] for l in self.layout]:
                self._build(params)
            self.key = key
        n = len(self.layout)
        host = (L.WnJob * n)()
        for i, (off, cnt, shape, r0, rows) in enumerate(self.layout):
            v, g = params[2 * i], params[2 * i + 1]
            j = host[i]
            j.v, j.g, j.w, j.inv = v.data_ptr(), g.data_ptr(), self.flat.data_ptr() + 4 * off, self.inv.data_ptr() + 4 * r0
            j.rows, j.cols, j.row0 = rows, cnt // rows, r0
            if backward:
                j.dw, j.dv, j.dg = dws[i].data_ptr(), dv.data_ptr() + 4 * off, dg.data_ptr() + 4 * r0
        nbytes = C.sizeof(L.WnJob) * n
        st = _stream()
        table = _upload_table(C.addressof(host), nbytes, _roundup(nbytes, 16), params[0].device, st)
        L.check(L.load().srk_weight_norm_group(table.data_ptr(), n, self.rows, 1 if backward else 0, st), "srk_weight_norm_group")

    def weights(self):
        params = []
        for c in self.convs:
            params += [c.weight_v, c.weight_g]
        _need_gpu(params[0])
        ws = _WnFn.apply(self, *params)
        for w, c in zip(ws, self.convs):
            w.__dict__["_srk_wn_proxy"] = True
            w.__dict__["_srk_pack_key"] = ("wn", id(c.weight_v))
        return list(ws)


# --------------------------------------------------------------------------------------------
# WDSR _Block_B with the 6F-channel intermediate kept on chip (csrc/pw_chain.hip)
# --------------------------------------------------------------------------------------------
_PW_OFF = _knob("SRK_NO_PW", "0") == "1"        # A/B knob: the block as three srk_conv2d launches (ConvChainFn)


def pw_ok(x, w1, w2):
    """Whether `conv1x1(w1) -> ReLU -> conv1x1(w2)` on NHWC `x` runs as ONE srk_pw_forward launch (16-bit storage,
    WDSR-B's shapes at n_feats 128 / 64: srk_pw_shape_ok)."""
    if _PW_OFF or x.dtype not in (torch.bfloat16, torch.float16) or x.numel() == 0:
        return False
    if w1.shape[2] != 1 or w2.shape[2] != 1 or w1.shape[1] != x.shape[3] or w2.shape[1] != w1.shape[0]:
        return False
    if x.numel() // x.shape[3] * max(x.shape[3], _roundup(w2.shape[0], 64)) * 2 >= _ADDR_LIMIT:
        return False
    return bool(L.load().srk_pw_shape_ok(int(w1.shape[1]), int(w1.shape[0]), _roundup(int(w2.shape[0]), 64)))


class PwPacked:
    __slots__ = ("fwd", "bwd", "cin", "chid", "cmid", "coutp")


def pw_pack(w1, b1, w2, b2, dtype, token=None):
    """fp32 [Chid][Cin][1][1] / [Cmid][Chid][1][1] (+ biases) -> the forward and backward slice streams of srk_pw_*.
    Weights with a stable identity (parameters, WeightNormGroup proxies) are served by / registered with the open PackGroup."""
    _need_gpu(w1)
    lib = L.load()
    ids = []
    for w in (w1, w2):
        ids.append(id(w) if isinstance(w, torch.nn.Parameter) else w.__dict__.get("_srk_pack_key"))
    group = _group_for(token) if all(i is not None for i in ids) else None
    key = (ids[0], ids[1], dtype)
    if group is not None:
        hit = group.lookup_pw(key)
        if hit is not None:
            return hit
    p = PwPacked()
    p.cin, p.chid, p.cmid = int(w1.shape[1]), int(w1.shape[0]), int(w2.shape[0])
    p.coutp = _roundup(p.cmid, 64)
    p.fwd = torch.empty(lib.srk_pw_pack_bytes(p.cin, p.chid, p.coutp, 0), dtype=torch.uint8, device=w1.device)
    p.bwd = torch.empty(lib.srk_pw_pack_bytes(p.cin, p.chid, p.coutp, 1), dtype=torch.uint8, device=w1.device)
    w1f, w2f = _f32c(w1.detach()), _f32c(w2.detach())
    b1f = None if b1 is None else _f32c(b1.detach())
    b2f = None if b2 is None else _f32c(b2.detach())
    a = L.PwPackArgs(w1=w1f.data_ptr(), b1=_ptr(b1f), w2=w2f.data_ptr(), b2=_ptr(b2f), Cin=p.cin, Chid=p.chid,
                     Cmid=p.cmid, CoutP=p.coutp, fwd=p.fwd.data_ptr(), bwd=p.bwd.data_ptr(), dtype=_DT[dtype])
    L.call("srk_pw_pack", a, _stream())
    if (group is not None and w1f.data_ptr() == w1.data_ptr() and w2f.data_ptr() == w2.data_ptr()
            and (b1 is None or b1f.data_ptr() == b1.data_ptr()) and (b2 is None or b2f.data_ptr() == b2.data_ptr())):
        group.add_pw(key, a, p, (w1, b1, w2, b2))
    return p


def pw_forward_raw(x, pk, out):
    n, h, wd, _ = x.shape
    L.call("srk_pw_forward", L.PwArgs(x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, P=n * h * wd, Cin=pk.cin, Chid=pk.chid, CoutP=pk.coutp,
                                      Cout=out.shape[3], wpk=pk.fwd.data_ptr(), out=out.data_ptr(), out_pitch=_pitch(out), out_coff=0,
                                      dtype=_DT[x.dtype]), _stream())
    return out


def pw_backward_raw(x, gz, pk, gx, *, res=None, h_out=None, gh_out=None):
    n, h, wd, _ = x.shape
    L.call("srk_pw_backward", L.PwBwdArgs(
        x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, gz=gz.data_ptr(), gz_pitch=_pitch(gz), gz_coff=0, Cz=gz.shape[3], P=n * h * wd,
        Cin=pk.cin, Chid=pk.chid, CoutP=pk.coutp, wpk=pk.bwd.data_ptr(), res=_ptr(res), res_pitch=0 if res is None else _pitch(res), res_coff=0,
        gx=gx.data_ptr(), gx_pitch=_pitch(gx), gx_coff=0, h_out=_ptr(h_out), gh_out=_ptr(gh_out), dtype=_DT[x.dtype]), _stream())
    return gx


def pw_wgrad_raw(x, gz, pk, w1_shape, w2_shape, want_b1=True, want_b2=True, defer=False):
    """dW1, db1, dW2, db2 of the pointwise pair from x and gz alone (srk_pw_wgrad: h and gh are re-computed tile by tile)."""
    n, h, wd, _ = x.shape
    P = n * h * wd
    dev = x.device
    nr = L.load().srk_pw_wgrad_ranges(P, pk.chid)
    scratch = torch.empty(nr * (pk.chid * (pk.cin + pk.coutp + 1) + pk.coutp), dtype=torch.float32, device=dev)
    dw1 = torch.empty(w1_shape, dtype=torch.float32, device=dev)
    dw2 = torch.empty(w2_shape, dtype=torch.float32, device=dev)
    db1 = torch.empty(pk.chid, dtype=torch.float32, device=dev) if want_b1 else None
    db2 = torch.empty(pk.cmid, dtype=torch.float32, device=dev) if want_b2 else None
    o2 = nr * pk.chid * pk.cin
    o3 = o2 + nr * pk.chid * pk.coutp
    o4 = o3 + nr * pk.chid
    a = L.PwWgradArgs(
        x=x.data_ptr(), x_pitch=_pitch(x), x_coff=0, gz=gz.data_ptr(), gz_pitch=_pitch(gz), gz_coff=0, Cz=gz.shape[3], P=P,
        Cin=pk.cin, Chid=pk.chid, Cmid=pk.cmid, CoutP=pk.coutp, wpk=pk.bwd.data_ptr(),
        dw1p=scratch.data_ptr(), dw2p=scratch.data_ptr() + 4 * o2, db1p=scratch.data_ptr() + 4 * o3,
        db2p=(scratch.data_ptr() + 4 * o4) if want_b2 else 0, nranges=nr,
        dw1=dw1.data_ptr(), db1=_ptr(db1), dw2=dw2.data_ptr(), db2=_ptr(db2), dtype=_DT[x.dtype])
    if defer and _WQ.enabled and not _PW_FIN_EACH and _arm_flush():
        # inside a backward pass: the slabs now, their sums with every other pointwise pair's in ONE launch when the pass's deferred
        # weight gradients are flushed (the results are read no earlier: WeightNormGroup's node flushes first)
        L.call("srk_pw_wgrad_partial", a, _stream())
        # (the STORAGES, not the tensors: AccumulateGrad adopts a gradient nobody else references, and clones -- now, unfilled -- one that is)
        _WQ.pwjobs.append(dict(a=a, keep=[scratch] + [t.untyped_storage() for t in (dw1, db1, dw2, db2) if t is not None]))
        _WQ.stream = torch.cuda.current_stream()
    else:
        L.call("srk_pw_wgrad", a, _stream())
    return dw1, db1, dw2, db2


# One finalize launch for ALL pointwise pairs of a backward pass (srk_pw_wgrad_finalize_group) is OFF by default: measured on two boxes
# (tools/ab_pw.sh, profiles/r5_ab_pw_b16.txt) it is -1 % ... +0.5 % against a finalize per pair -- each pair's 31.5 MB of slabs are still in the
# Infinity Cache when its own finalize reads them right behind the kernel that wrote them; all sixteen read at the end of the pass (504 MB) come
# from HBM.  SRK_DEBUG=1 SRK_PW_GROUP_FIN=1 selects the grouped form.
_PW_FIN_EACH = _knob("SRK_PW_GROUP_FIN", "0") != "1"
_PW_WG_OFF = _knob("SRK_NO_PW_WGRAD", "0") == "1"      # A/B knob: h / gh through HBM + the two 1x1 weight-gradient GEMMs


class WdsrBlockBFn(torch.autograd.Function):
    """WDSR _Block_B (models/wdsr.py:30-51): out = conv3x3(conv1x1(relu(conv1x1(x)))) * res_scale + x.

    forward : srk_pw_forward (both pointwise convs, the 6F-channel tensor never leaves the chip) + one srk_conv2d (3x3, * scale, + x)
    backward: dgrad 3x3 (* scale), srk_pw_backward (re-computes the ReLU mask, + g of the skip connection; also leaves h and gh
              for the pointwise weight gradients), three weight gradients."""

    @staticmethod
    def forward(ctx, x, scale, w1, b1, w2, b2, w3, b3):
        _need_gpu(x)
        n, h, wd, cp = x.shape
        dt = x.dtype
        pk = pw_pack(w1, b1, w2, b2, dt)
        ctx.pg = _tok()
        z = torch.empty((n, h, wd, pad16(w2.shape[0])), dtype=dt, device=x.device)
        pw_forward_raw(x, pk, z)
        out = torch.empty_like(x)
        conv_raw(z, pack_conv(w3, b3, dt), N=n, H=h, W=wd, Cin=z.shape[3], Cout=cp, out=out, scale=scale, res=x)
        ctx.save_for_backward(x, z, w1, w2, w3)
        ctx.pk, ctx.scale = pk, scale
        ctx.wb = (w1, b1, w2, b2, w3, b3)
        ctx.pg = _tok()
        return out

    @staticmethod
    def backward(ctx, g):
        x, z, w1, w2, w3 = ctx.saved_tensors
        scale = ctx.scale
        _, b1, _, b2, _, b3 = ctx.wb
        g = g.contiguous()
        n, h, wd, cp = x.shape
        dt = x.dtype
        # a pack that lives in the model's PackGroup is refreshed in place by later forwards: take it through the token like
        # every dgrad pack (re-packed from the current weights if the window has moved on); a private pack is immutable
        pk = ctx.pk if ctx.pg is None else pw_pack(ctx.wb[0], b1, ctx.wb[2], b2, dt, token=ctx.pg)
        gz = torch.empty_like(z)
        conv_raw(g, pack_conv(w3, None, dt, dgrad=True, token=ctx.pg), N=n, H=h, W=wd, Cin=cp, Cout=z.shape[3], out=gz, scale=scale, use_bias=False)
        gw3, gb3 = wgrad(z, g, wparam=ctx.wb[4], bparam=b3, N=n, H=h, W=wd, Cin=z.shape[3], Cout=cp, k=3, w_shape=tuple(w3.shape),
                         scale=scale, want_bias=b3 is not None)
        gx = torch.empty_like(x)
        if _PW_WG_OFF:
            hid = torch.empty((n, h, wd, pk.chid), dtype=dt, device=x.device)
            ghid = torch.empty_like(hid)
            pw_backward_raw(x, gz, pk, gx, res=g, h_out=hid, gh_out=ghid)
            gw2, gb2 = wgrad(hid, gz, wparam=ctx.wb[2], bparam=b2, N=n, H=h, W=wd, Cin=pk.chid, Cout=gz.shape[3], k=1, w_shape=tuple(w2.shape),
                             want_bias=b2 is not None)
            gw1, gb1 = wgrad(x, ghid, wparam=ctx.wb[0], bparam=b1, N=n, H=h, W=wd, Cin=cp, Cout=pk.chid, k=1, w_shape=tuple(w1.shape),
                             want_bias=b1 is not None)
            return gx, None, gw1, gb1, gw2, gb2, gw3, gb3
        pw_backward_raw(x, gz, pk, gx, res=g)
        # deferred finalize only where autograd will ADOPT the (still unfilled) results: no existing .grad to accumulate into, no hooks
        can_defer = all((sl := _grad_slot(p_, sh_)) is not None and sl[0] == "new"
                        for p_, sh_ in ((ctx.wb[0], tuple(w1.shape)), (b1, (w1.shape[0],)), (ctx.wb[2], tuple(w2.shape)), (b2, (w2.shape[0],))))
        gw1, gb1, gw2, gb2 = pw_wgrad_raw(x, gz, pk, tuple(w1.shape), tuple(w2.shape), want_b1=b1 is not None, want_b2=b2 is not None, defer=can_defer)
        return gx, None, gw1, gb1, gw2, gb2, gw3, gb3


def wdsr_block_b(x, convs, scale=1.0):
    """convs = [(w1, b1), (w2, b2), (w3, b3)] of a _Block_B; the fused form when the shapes allow, else the conv chain."""
    (w1, b1), (w2, b2), (w3, b3) = convs
    if pw_ok(x, w1, w2) and w3.shape[2] == 3 and w3.shape[0] == x.shape[3]:
        return WdsrBlockBFn.apply(x, float(scale), w1, b1, w2, b2, w3, b3)
    return conv_chain(x, convs, [True, False, False], scale=scale)



class _CAHint:
    """Link between consecutive RCABs of one forward pass (small batches only): block k leaves its conv output `t` here; block
    k+1 -- whose backward produces the gradient block k receives -- pools t * gradient while that gradient leaves its
    dgrad launch (srk_conv_pair pool / pool_aux), which is the pooling pass block k's channel-attention backward starts
    with.  Block k uses the sums only for the very tensor they were formed from (same storage, untouched since)."""
    __slots__ = ("t", "out_ref", "gsum", "g_ptr", "g_ver", "__weakref__")

    def __init__(self, t, out):
        self.t, self.out_ref = t, weakref.ref(out)
        self.gsum = self.g_ptr = self.g_ver = None


class _LazyApply:
    """The pending last step of an RCAB (out = t * s + x with s from the squeeze/excite MLP on `sums`): everything the next
    block's launch needs to perform it, and the buffers (`out`, `s`, `z`) it fills."""
    __slots__ = ("t", "x", "sums", "w", "s", "z", "out")

    def __init__(self, t, x, sums, w, s, z, out):
        self.t, self.x, self.sums, self.w, self.s, self.z, self.out = t, x, sums, w, s, z, out


def _fwd_state():
    """Per-thread hand-over slots between consecutive RCABFn.forward calls (`ca_hint`: the previous block's _CAHint,
    `lazy`: its pending _LazyApply).  Thread-local: two models stepping on two Python threads never see each other's."""
    st = _TLS
    if not hasattr(st, "ca_hint"):
        st.ca_hint = st.lazy = None
    return st


#: launches of srk_conv_pair by ca_mode (0 plain, 1 CALayer backward on the way in, 2 CALayer forward on the way in):
#: tests assert from it that a model run really took the fused flavours
PAIR_LAUNCHES = [0, 0, 0]


class RCABFn(torch.autograd.Function):
    """RCAB (models/rcan.py:33-55): conv -> ReLU -> conv -> CALayer (rcan.py:10-29), += x.

    forward : 2 convs, srk_ca_pool (sum over HxW), srk_ca_apply (MLP + sigmoid scale + residual)
    backward: srk_ca_pool(t*g), srk_ca_bwd_apply, dgrad2 (ReLU mask), dgrad1 (+g), 2 wgrads."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, cw1, cb1, cw2, cb2, lazy_in=None, lazy_out=False):
        """lazy_in / lazy_out (rcab_chain, small batches): with lazy_out the block leaves its last step -- out = t * s + x,
        the srk_ca_apply launch -- to the NEXT block's conv launch, which performs it on its input tile as it loads it
        (srk_conv_pair ca_mode 2) and fills `out`, `s` and `z`; `lazy_in` is that pending step of the previous block."""
        _need_gpu(x)
        n, h, wd, cp = x.shape
        dt = x.dtype
        c = w2.shape[0]
        cr = cw1.shape[0]
        y1 = torch.empty_like(x)
        t = torch.empty_like(x)
        paired = pair_ok(x, w1, w2)
        assert paired or (lazy_in is None and not lazy_out), "the lazy channel-attention protocol is for the paired launches"
        if paired:
            # small batch: both convs in one launch, which also leaves the per-tile channel sums of t (the pooling pass)
            ns = L.load().srk_conv_pair_tiles(1, h, wd)
            sums = torch.empty((n, ns, cp), dtype=torch.float32, device=x.device)
            if lazy_in is None:
                conv_pair_raw(x, pack_conv(w1, b1, dt), pack_conv(w2, b2, dt), out=t, relu_mid=True, mid=y1, pool=sums)
            else:       # x = lazy_in.t * s + lazy_in.x is formed on the way in and stored to x
                conv_pair_raw(lazy_in.t, pack_conv(w1, b1, dt), pack_conv(w2, b2, dt), out=t, relu_mid=True, mid=y1, pool=sums,
                              ca_fwd=dict(x2=lazy_in.x, sums=lazy_in.sums, w1=lazy_in.w[0], b1=lazy_in.w[1], w2=lazy_in.w[2],
                                          b2=lazy_in.w[3], s_out=lazy_in.s, z_out=lazy_in.z), xo=x)
        else:
            conv_raw(x, pack_conv(w1, b1, dt), N=n, H=h, W=wd, Cin=cp, Cout=cp, out=y1, relu=True, relu_bits="want")
            ctx.bits = y1.__dict__.pop("_srk_bits", None)
            conv_raw(y1, pack_conv(w2, b2, dt), N=n, H=h, W=wd, Cin=cp, Cout=cp, out=t)
            ns = L.load().srk_ca_splits(n, h * wd)
            sums = torch.empty((n, ns, cp), dtype=torch.float32, device=x.device)      # per-block partials: nothing to zero
            L.call("srk_ca_pool", L.CaPoolArgs(t=t.data_ptr(), t_pitch=cp, t_coff=0, u=0, u_pitch=0, u_coff=0,
                                               sums=sums.data_ptr(), N=n, HW=h * wd, C=cp, dtype=_DT[dt]), _stream())
        s = torch.empty((n, cp), dtype=torch.float32, device=x.device)
        z = torch.empty((n, cr), dtype=torch.float32, device=x.device)
        out = torch.empty_like(x)
        w1f, b1f, w2f, b2f = _ca_params(cw1, cb1, cw2, cb2, cp)
        fst = _fwd_state()
        fst.lazy = None
        if lazy_out:
            fst.lazy = _LazyApply(t, x, sums, (w1f, b1f, w2f, b2f), s, z, out)
        else:
            L.call("srk_ca_apply", L.CaApplyArgs(
                t=t.data_ptr(), t_pitch=cp, t_coff=0, res=x.data_ptr(), res_pitch=cp, res_coff=0, sums=sums.data_ptr(),
                w1=w1f.data_ptr(), b1=b1f.data_ptr(), w2=w2f.data_ptr(), b2=b2f.data_ptr(),
                s_out=s.data_ptr(), z_out=z.data_ptr(), out=out.data_ptr(), out_pitch=cp, out_coff=0,
                N=n, HW=h * wd, C=cp, Cr=cr, dtype=_DT[dt], sums_rows=ns), _stream())
        ctx.save_for_backward(x, y1, t, sums, s, z, w1, w2, cw1, cw2)
        ctx.wb = (w1, b1, w2, b2)
        ctx.ca = (cw1, cb1, cw2, cb2)
        ctx.pg = _tok()
        prev, ctx.hint_in, ctx.hint_out = fst.ca_hint, None, None
        if paired:
            if prev is not None and prev.out_ref() is x and prev.t.shape == x.shape:
                ctx.hint_in = prev                           # x is the previous RCAB's output
            ctx.hint_out = fst.ca_hint = _CAHint(t, out)
        else:
            fst.ca_hint = None
        return out

    @staticmethod
    def backward(ctx, g):
        x, y1, t, sums, s, z, w1, w2, cw1, cw2 = ctx.saved_tensors
        g = g.contiguous()
        n, h, wd, cp = x.shape
        dt = x.dtype
        c, cr = w2.shape[0], cw1.shape[0]
        dev = x.device
        ho = ctx.hint_out
        if ho is not None and ho.gsum is not None and ho.g_ptr == g.data_ptr() and ho.g_ver == g._version and ho.gsum.shape[0] == n:
            gsum = ho.gsum                                   # sums of t * g formed by the launch that produced g
            gs_rows = gsum.shape[1]
        else:
            gs_rows = L.load().srk_ca_splits(n, h * wd)
            gsum = torch.empty((n, gs_rows, cp), dtype=torch.float32, device=dev)
            L.call("srk_ca_pool", L.CaPoolArgs(t=t.data_ptr(), t_pitch=cp, t_coff=0, u=g.data_ptr(), u_pitch=_pitch(g), u_coff=0,
                                               sums=gsum.data_ptr(), N=n, HW=h * wd, C=cp, dtype=_DT[dt]), _stream())
        w1f, _, w2f, _ = _ca_params(cw1, None, cw2, None, cp)
        # per-sample parameter-gradient contributions [N][dW1 | db1 | dW2 | db2], summed over n below (one reduction
        # instead of a zero-fill + float atomics + per-tensor copies)
        per = torch.empty((max(n, 1), 2 * cr * cp + cr + cp), dtype=torch.float32, device=dev)
        if n == 0:
            per.zero_()
        dw1, db1 = per[0, :cr * cp], per[0, cr * cp:cr * cp + cr]
        dw2, db2 = per[0, cr * cp + cr:2 * cr * cp + cr], per[0, 2 * cr * cp + cr:]
        gt = torch.empty_like(x)
        paired = pair_ok(g, w1, w2)
        fuse_ca = paired and cp == 64 and cr <= 8 and gs_rows <= 64 and sums.shape[1] <= 64 and not _CA_UNFUSED
        if not fuse_ca:
            L.call("srk_ca_bwd_apply", L.CaBwdArgs(
                g=g.data_ptr(), g_pitch=_pitch(g), g_coff=0, gsum=gsum.data_ptr(), sums=sums.data_ptr(), s=s.data_ptr(),
                z=z.data_ptr(), w1=w1f.data_ptr(), w2=w2f.data_ptr(), dw1=dw1.data_ptr(), db1=db1.data_ptr(),
                dw2=dw2.data_ptr(), db2=db2.data_ptr(), gt=gt.data_ptr(), gt_pitch=cp, gt_coff=0,
                N=n, HW=h * wd, C=cp, Cr=cr, dtype=_DT[dt], sums_rows=sums.shape[1], gsum_rows=gs_rows), _stream())
        g1 = torch.empty_like(x)
        gx = torch.empty_like(x)
        if paired:
            # ONE launch: the CALayer backward on the way in (gt = g*s + dmean, parameter-gradient slots), dgrad of conv 2
            # with the ReLU mask, dgrad of conv 1, + g; gt and g1 are stored for the two weight gradients.  If the previous
            # RCAB left its t, the launch also pools t * gx for that block's backward.
            hi = ctx.hint_in
            pool = None
            if hi is not None and hi.t.shape == gx.shape and hi.t.dtype == dt:
                pool = torch.empty((n, L.load().srk_conv_pair_tiles(1, h, wd), cp), dtype=torch.float32, device=dev)
            conv_pair_raw(g if fuse_ca else gt, pack_conv(w2, None, dt, dgrad=True, token=ctx.pg),
                          pack_conv(w1, None, dt, dgrad=True, token=ctx.pg),
                          out=gx, mask=y1, mid=g1, res=g, use_bias=False, pool=pool, pool_aux=None if pool is None else hi.t,
                          ca_bwd=dict(gsum=gsum, sums=sums, s=s, z=z, w1=w1f, w2=w2f, slots=per) if fuse_ca else None,
                          xo=gt if fuse_ca else None)
            if pool is not None:
                hi.gsum, hi.g_ptr, hi.g_ver = pool, gx.data_ptr(), gx._version
        else:
            conv_raw(gt, pack_conv(w2, None, dt, dgrad=True, token=ctx.pg), N=n, H=h, W=wd, Cin=cp, Cout=cp, out=g1, mask=y1,
                     mask_bits=getattr(ctx, "bits", None), use_bias=False)
        gw2, gb2 = wgrad(y1, gt, wparam=ctx.wb[2], bparam=ctx.wb[3], N=n, H=h, W=wd, Cin=cp, Cout=cp, k=w2.shape[2],
                         w_shape=tuple(w2.shape), want_bias=ctx.wb[3] is not None)
        gw1, gb1 = wgrad(x, g1, wparam=ctx.wb[0], bparam=ctx.wb[1], N=n, H=h, W=wd, Cin=cp, Cout=cp, k=w1.shape[2],
                         w_shape=tuple(w1.shape), want_bias=ctx.wb[1] is not None)
        if not paired:
            conv_raw(g1, pack_conv(w1, None, dt, dgrad=True, token=ctx.pg), N=n, H=h, W=wd, Cin=cp, Cout=cp, out=gx, res=g, use_bias=False)
        # The sum of the slots over n.  No padding channels (C == Cp): deferred, ONE launch at the end of the backward pass
        # sums the slots of every RCAB (ops.defer_rowsum).  (Summing inside this launch -- last block by ticket -- was
        # measured: the device-scope release it needs writes back the L2 of every XCD and costs what a reduction launch costs.)
        if c == cp:
            o1, o2, o3 = cr * cp, cr * cp + cr, 2 * cr * cp + cr
            outs = defer_rowsum(per, (ctx.ca[0], ctx.ca[1], ctx.ca[2], ctx.ca[3]),
                                (((cr, c, 1, 1), 0), ((cr,), o1), ((c, cr, 1, 1), o2), ((c,), o3)))
            if outs is not None:
                return (gx, gw1, gb1, gw2, gb2, *outs, None, None)
        tot = per.sum(0)
        dw1, db1 = tot[:cr * cp], tot[cr * cp:cr * cp + cr]
        dw2, db2 = tot[cr * cp + cr:2 * cr * cp + cr], tot[2 * cr * cp + cr:]
        # un-pad the CA parameter gradients (rows/cols beyond the real C are zero); views of `tot`, no copies when C == Cp
        gcw1 = dw1.view(cr, cp)[:, :c].reshape(cr, c, 1, 1)
        gcw2 = dw2.view(cp, cr)[:c].reshape(c, cr, 1, 1)
        return gx, gw1, gb1, gw2, gb2, gcw1, db1, gcw2, db2[:c], None, None


def _ca_params(cw1, cb1, cw2, cb2, cp):
    """fp32 [Cr][Cp] / [Cp][Cr] views of conv_du.{0,2} padded to the activation's channel count."""
    cr, c = cw1.shape[0], cw1.shape[1]
    w1 = _f32c(cw1).view(cr, c)
    w2 = _f32c(cw2).view(c, cr)
    b1 = None if cb1 is None else _f32c(cb1)
    b2 = None if cb2 is None else _f32c(cb2)
    if cp != c:
        w1 = torch.nn.functional.pad(w1, (0, cp - c)).contiguous()
        w2 = torch.nn.functional.pad(w2, (0, 0, 0, cp - c)).contiguous()
        if b2 is not None:
            b2 = torch.nn.functional.pad(b2, (0, cp - c)).contiguous()
    return w1, b1, w2, b2


def rcab(x, w1, b1, w2, b2, cw1, cb1, cw2, cb2):
    return RCABFn.apply(x, w1, b1, w2, b2, cw1, cb1, cw2, cb2)


def rcab_chain(x, blocks):
    """A run of RCABs (a residual group's body, models/rcan.py:59-74); blocks[i] = the nine parameters of `rcab`.

    At small batches (pair_ok) every block is ONE launch forward: the conv pair also pools its output, and the channel
    attention's last step of block i (t * s + x) is performed by block i+1's launch on its input tile; only the last block
    runs srk_ca_apply.  Larger batches: the blocks one by one, as `rcab`."""
    if not blocks:
        return x
    fst = _fwd_state()
    w1, w2 = blocks[0][0], blocks[0][2]
    lazy_ok = (not _CA_UNFUSED and pair_ok(x, w1, w2) and x.shape[3] == 64 and L.load().srk_conv_pair_tiles(1, x.shape[1], x.shape[2]) <= 64
               and all(tuple(b[0].shape) == (64, 64, 3, 3) and tuple(b[2].shape) == (64, 64, 3, 3) and b[5] is not None and b[7] is not None
                       and b[4].shape[1] == 64 and b[4].shape[0] <= 8 for b in blocks))
    lazy = None
    for i, b in enumerate(blocks):
        last = i == len(blocks) - 1
        x = RCABFn.apply(x, *b, lazy, lazy_ok and not last)
        lazy = fst.lazy if (lazy_ok and not last) else None
        fst.lazy = None
    return x


class RDBFn(torch.autograd.Function):
    """Residual dense block (models/rdn.py:9-40): C x [relu(conv3x3(feat)) appended to feat], 1x1 LFF, + x.

    The reference re-copies the growing tensor with torch.cat at every layer (rdn.py:21); here all
    layers write their G channels into a slice of ONE pre-allocated [N,H,W,G0+C*G] NHWC buffer and read
    the prefix of it (pitch = full width).  Backward walks the chain in reverse on a gradient buffer
    of the same shape: every dgrad accumulates into the prefix (res = out) and applies the ReLU mask
    of the newest slice it completes."""

    @staticmethod
    def forward(ctx, x, dest, *params):
        """dest: None, or (SliceBuffer, index): the block's output is WRITTEN into channel slice `index` of that pre-allocated
        buffer (RDN's global feature fusion reads the concatenation of all block outputs, rdn.py:108: no torch.cat copy) and
        the returned tensor is that slice (a pitched NHWC view)."""
        _need_gpu(x)
        n, h, wd, g0 = x.shape
        nconv = (len(params) - 2) // 2
        ws, bs = params[0:2 * nconv:2], params[1:2 * nconv:2]
        wl, bl = params[-2], params[-1]
        g = ws[0].shape[0]
        assert g0 % 16 == 0 and g % 16 == 0, "dense-block widths must be multiples of 16"
        ctot = g0 + nconv * g
        feat = torch.empty((n, h, wd, ctot), dtype=x.dtype, device=x.device)
        feat[..., :g0].copy_(x)
        for c in range(nconv):
            cin = g0 + c * g
            conv_raw(feat[..., :cin], pack_conv(ws[c], bs[c], x.dtype), N=n, H=h, W=wd, Cin=cin, Cout=g,
                     out=feat[..., cin:cin + g], relu=True)
        out = torch.empty_like(x) if dest is None else dest[0].slice(dest[1], x)
        conv_raw(feat, pack_conv(wl, bl, x.dtype), N=n, H=h, W=wd, Cin=ctot, Cout=g0, out=out, res=x)
        ctx.save_for_backward(feat, *ws, wl)
        ctx.cfg = (nconv, g0, g)
        ctx.wb = (tuple(ws), tuple(bs))
        ctx.pg = _tok()
        return out

    @staticmethod
    def backward(ctx, gout):
        nconv, g0, g = ctx.cfg
        feat = ctx.saved_tensors[0]
        ws, wl = ctx.saved_tensors[1:1 + nconv], ctx.saved_tensors[-1]
        gout = gout.contiguous()
        n, h, wd, ctot = feat.shape
        dt = feat.dtype
        grads = [None] * (2 * nconv + 2)
        gwl, gbl = wgrad_raw(feat, gout, N=n, H=h, W=wd, Cin=ctot, Cout=g0, k=1, w_shape=tuple(wl.shape))
        grads[-2], grads[-1] = gwl, gbl
        gfeat = torch.empty_like(feat)
        # LFF dgrad fills the whole gradient buffer; the top slice (output of the last dense conv) gets its ReLU mask
        conv_raw(gout, pack_conv(wl, None, dt, dgrad=True, token=ctx.pg), N=n, H=h, W=wd, Cin=g0, Cout=ctot, out=gfeat,
                 mask=feat, mask_from=ctot - g, use_bias=False)
        for c in range(nconv - 1, -1, -1):
            cin = g0 + c * g
            dy = gfeat[..., cin:cin + g]
            # (slice c of gfeat is final here: the remaining dgrads only touch the channels below it, so the job can wait)
            gw, gb = wgrad(feat[..., :cin], dy, wparam=ctx.wb[0][c], bparam=ctx.wb[1][c], N=n, H=h, W=wd, Cin=cin, Cout=g,
                           k=ws[c].shape[2], w_shape=tuple(ws[c].shape), want_bias=ctx.wb[1][c] is not None)
            grads[2 * c], grads[2 * c + 1] = gw, gb
            pref = gfeat[..., :cin]
            conv_raw(dy, pack_conv(ws[c], None, dt, dgrad=True, token=ctx.pg), N=n, H=h, W=wd, Cin=g, Cout=cin, out=pref, res=pref,
                     mask=feat[..., :cin] if c > 0 else None, mask_from=cin - g if c > 0 else 0, use_bias=False)
        gx = gfeat[..., :g0] + gout
        return (gx, None, *grads)


_SLICE_GACC = _knob("SRK_NO_SLICE_GACC", "0") != "1"      # A/B knob: autograd's own sums of the concatenations' gradient slices


class SliceBuffer:
    """One [N, H, W, count * C] NHWC buffer whose channel slices are the outputs of `count` blocks (RDN: the D residual dense
    blocks feeding the global feature fusion, models/rdn.py:99-108; D-DBPN: the HR feature maps whose growing concatenations the
    down units read, ddbpn.py:116-134).  A plain Python object, so autograd sees neither the buffer nor the in-place slice writes:
    every block returns its slice as a fresh output tensor, `ConcatSlicesFn` returns the buffer (or a prefix of it) as the
    concatenation and hands each block its slice of the gradient (views, no copy either way).

    With `accumulate_grads` (several concatenations of growing prefixes: D-DBPN) the gradients of all consumers meet in ONE
    buffer of the same shape: the consumer of the longest prefix runs first in backward and its gradient becomes the buffer, every
    later consumer adds its gradient to its prefix -- a 1x1 conv does so in its own data-gradient launch (`res` = `out` = the
    prefix: ConvFn) -- and a part's slice is handed to autograd by the LAST consumer that covers it.  Autograd itself would add
    the (count - i) slices of part i one strided launch at a time (15 launches over 37.7 MB each per D-DBPN step at the
    reference's batch).  The order (longest prefix first) is what the data flow forces -- part k - 1 is produced from the output of
    the consumer of prefix k - 1 -- and is checked: a consumer arriving out of order raises."""

    def __init__(self, count, accumulate_grads=False):
        self.count, self.buf = int(count), None
        self.accumulate = bool(accumulate_grads) and _SLICE_GACC
        self.ks = []            # prefix lengths of the concatenations taken in this forward
        self.gbuf, self.last_k = None, None

    def alloc(self, n, h, w, c, dtype, device):
        self.buf = torch.empty((n, h, w, self.count * c), dtype=dtype, device=device)
        self.c = c
        return self

    def slice(self, i, like=None):
        if self.buf is None:
            n, h, w, c = like.shape
            self.alloc(n, h, w, c, like.dtype, like.device)
        c = self.c if like is None else like.shape[3]
        return self.buf[..., i * c:(i + 1) * c]

    # -- gradient side (accumulate_grads) --
    def grad_dest(self, k):
        """Where the consumer of prefix `k` writes its data gradient: (view of the gradient buffer, add to what is there?)."""
        if self.gbuf is None:
            if k != max(self.ks):
                raise RuntimeError(f"SliceBuffer: the consumer of prefix {k} runs its backward before the one of prefix {max(self.ks)}")
            self.gbuf = torch.empty_like(self.buf)
            self.last_k = None
            first = True
        else:
            first = False
        return self.gbuf[..., :k * self.c], not first

    def deliver(self, k, g):
        """Called by the concatenation's backward with the consumer's gradient `g` of prefix `k`: makes sure it is in the buffer
        and returns the per-part gradients this consumer hands to autograd (None for parts a later consumer covers)."""
        c = self.c
        if self.gbuf is not None and g.data_ptr() == self.gbuf.data_ptr():
            pass                                                    # written / added in place by the consumer's own launch
        elif self.gbuf is None:
            if k != max(self.ks):
                raise RuntimeError(f"SliceBuffer: the consumer of prefix {k} runs its backward before the one of prefix {max(self.ks)}")
            if k == self.count and g.is_contiguous():
                self.gbuf = g                                       # the full-width gradient IS the buffer from here on
            else:
                self.gbuf = torch.zeros_like(self.buf)
                self.gbuf[..., :k * c].copy_(g)
            self.last_k = None
        else:
            self.gbuf[..., :k * c].add_(g)
        if self.last_k is not None and k >= self.last_k:
            raise RuntimeError(f"SliceBuffer: consumers must run their backward longest prefix first (prefix {k} after {self.last_k})")
        self.last_k = k
        below = [q for q in self.ks if q < k]
        lo = max(below) if below else 0
        out = [self.gbuf[..., i * c:(i + 1) * c] if i >= lo else None for i in range(k)]
        if not below:                                               # the last consumer: the next backward pass starts afresh
            self.gbuf, self.last_k = None, None
        return out


class ConcatSlicesFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder, *parts):
        ctx.c = c = parts[0].shape[3]
        ctx.holder, ctx.k = holder, len(parts)
        for i, p_ in enumerate(parts):
            assert p_.numel() == 0 or p_.data_ptr() == holder.buf.data_ptr() + i * ctx.c * p_.element_size(), \
                "part %d is not slice %d of the buffer" % (i, i)
        if holder.accumulate and len(parts) not in holder.ks:
            holder.ks.append(len(parts))
        if len(parts) == holder.count:
            out = holder.buf.view(holder.buf.shape)
        else:
            out = holder.buf[..., :len(parts) * c]          # a prefix: a channel-slice view the convs read with the buffer's pitch
        return out

    @staticmethod
    def backward(ctx, g):
        c = ctx.c
        if ctx.holder.accumulate:
            return (None, *ctx.holder.deliver(ctx.k, g))
        return (None, *[g[..., i * c:(i + 1) * c] for i in range(g.shape[3] // c)])


def concat_slices(holder, parts):
    """torch.cat(parts, dim=3) for parts that already ARE the first consecutive channel slices of `holder` (no copy): all of them
    (RDN's global fusion input, rdn.py:108) or a prefix (D-DBPN's growing concatenations, ddbpn.py:116-131)."""
    out = ConcatSlicesFn.apply(holder, *parts)
    if holder.accumulate:
        out.__dict__["_srk_gacc"] = (holder, len(parts))    # a conv that consumes it adds its data gradient in place (ConvFn)
    return out


class AddIntoFn(torch.autograd.Function):
    """slice `index` of a SliceBuffer = a + b (the `a_0.add(a_1)` that ends a D-DBPN projection unit, ddbpn.py:62-64, lands in
    its slot of the later concatenations).  Written by srk_chan_apply through the raw address, like every slice write: autograd
    sees a fresh output tensor, and the tensors other units saved of the same buffer keep their version."""

    @staticmethod
    def forward(ctx, a, b, holder, index):
        return chan_apply(a.contiguous(), y=b.contiguous(), out=holder.slice(index, a))

    @staticmethod
    def backward(ctx, g):
        return g, g, None, None


def add_into(a, b, dest):
    """dest: None (a plain sum) or (SliceBuffer, index)."""
    return AddIntoFn.apply(a, b, dest[0], dest[1]) if dest is not None else a.add(b)


def rdb(x, convs, lff, dest=None):
    flat = []
    for w, b in convs:
        flat += [w, b]
    return RDBFn.apply(x, dest, *flat, lff[0], lff[1])


# --------------------------------------------------------------------------------------------
# op families that live in their own modules (round 6): everything they define is part of this namespace, as before
# --------------------------------------------------------------------------------------------
from .ops_norm import *      # noqa: E402,F401,F403  BatchNorm2d, PReLU, per-channel statistics
from .ops_proj import *      # noqa: E402,F401,F403  unfold / fold, projection and general strided convs
from .ops_proj import _proj_launch      # noqa: E402,F401  (bench.py / tools/microbench_proj.py time the raw launches)
from .ops_metrics import *   # noqa: E402,F401,F403  PSNR / SSIM reductions, L1 loss
