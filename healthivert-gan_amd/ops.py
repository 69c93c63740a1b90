"""Python-side operator wrappers over the C ABI (include/hvgan.h).

Plumbing only: torch supplies device memory and the current HIP stream; all arithmetic happens in
libhvgan.so.  Activations are NHWC fp32 views (`Act`) with an explicit channel stride/offset so
convolutions read and write channel slices of concat buffers in place.
"""
import ctypes
import os

import torch

from . import lib as _lib
from .lib import ACT, NORM, F16, F32, ptr, stream

_PRECISION = {'fp32': F32, 'f32': F32, 'fp16': F16, 'f16': F16}


_TIMER = None


def timer():
    """The active per-launch kernel timer (profiler.KernelTimer) or None."""
    return _TIMER


def set_timer(t):
    """bench.py's live kernel timer (profiler.KernelTimer) or None."""
    global _TIMER
    _TIMER = t


def default_precision():
    """HV_PRECISION=fp32 (exact fp32 MFMA, parity mode) | fp16 (fp16 MFMA operands, fp32 accumulate)."""
    return _PRECISION[os.environ.get('HV_PRECISION', 'fp32').lower()]


def precision_id(p):
    if p is None:
        return default_precision()
    if isinstance(p, str):
        return _PRECISION[p.lower()]
    return int(p)


def rup(v, m):
    return (v + m - 1) // m * m


def cpad(C):
    """Channel stride of an activation buffer with C channels: a multiple of 4 (16 bytes of fp32 / 8 bytes of fp16 per lane), and of 8
    beyond 16 channels so that the fp16 kernels that stage 16-byte items (wgrad_tr_kernel) take the 33- and 65-channel concat buffers too."""
    return rup(C, 8) if C > 16 else rup(C, 4)


def bridge_grad_scale(precision):
    """Power-of-two gradient scale of the nn.Module autograd bridges (`loss.backward()` on module outputs) in the fp16 storage mode: the
    activation gradients of these networks are 1e-5 .. 1e-7 -- subnormal or zero in fp16 -- so the incoming seeds are multiplied by S, the
    backward is linear in them, and the parameter gradients / the input gradient are multiplied by 1/S afterwards (exact).  1 in the fp32 mode."""
    if precision_id(precision) != F16:
        return 1.0
    s = float(os.environ.get('HV_GRAD_SCALE', '8192'))
    return s if s > 0 else 1.0


def scale_inplace(t, factor):
    """t *= factor on the current stream (fp32 tensor)."""
    if factor != 1.0 and t is not None and t.numel():
        _lib.get().call('hv_affine', ptr(t), ptr(t), ctypes.c_longlong(t.numel()), ctypes.c_float(factor), ctypes.c_float(0.0), stream())


def storage_dtype(precision):
    """Element type of the activation / gradient tensors INSIDE a network for a compute precision: the fp16 mode stores them as
    fp16 (they are rounded to fp16 as MFMA operands anyway: half the HBM / L2 bytes, no conversions in the staging code), the
    exact-fp32 parity mode as fp32.  Image-level tensors of the reference API and the attention score matrices are always fp32."""
    return torch.float16 if precision_id(precision) == F16 else torch.float32


class Act:
    """NHWC activation view: tensor [B,H,W,ld] (fp32, or fp16 in the fp16 storage mode), channels [coff, coff+C)."""
    __slots__ = ('t', 'B', 'H', 'W', 'C', 'ld', 'coff', 'f16')

    def __init__(self, t, C=None, coff=0):
        assert t.dim() == 4 and t.dtype in (torch.float32, torch.float16) and t.is_contiguous(), (t.shape, t.dtype)
        _lib.require_gpu(t)
        self.t = t
        self.f16 = int(t.dtype == torch.float16)
        self.B, self.H, self.W, self.ld = t.shape
        self.coff = coff
        self.C = self.ld - coff if C is None else C
        assert 0 < self.C and self.coff + self.C <= self.ld

    @staticmethod
    def empty(B, H, W, C, device, ld=None, zero=False, dtype=torch.float32):
        ld = C if ld is None else ld
        t = (torch.zeros if zero else torch.empty)(B, H, W, ld, dtype=dtype, device=device)
        return Act(t, C, 0)

    def slice(self, coff, C):
        return Act(self.t, C, self.coff + coff)

    def like(self, zero=False):
        return Act.empty(self.B, self.H, self.W, self.ld, self.t.device, zero=zero, dtype=self.t.dtype).slice(self.coff, self.C) \
            if (self.coff or self.C != self.ld) else Act.empty(self.B, self.H, self.W, self.C, self.t.device, zero=zero, dtype=self.t.dtype)

    @property
    def npix(self):
        return self.B * self.H * self.W

    def nchw(self):
        """Copy out as a (B,C,H,W) tensor (API edge).  C == 1 is a free view."""
        if self.C == 1 and self.ld == 1 and not self.f16:
            return self.t.view(self.B, 1, self.H, self.W)
        out = torch.empty(self.B, self.C, self.H, self.W, dtype=torch.float32, device=self.t.device)
        L = _lib.get()
        L.call('hv_nhwc_to_nchw', ptr(self.t), self.f16, ptr(out), self.B, self.C, self.H, self.W, self.ld, self.coff, 0, stream())
        return out


def from_nchw(x, CP=None, dtype=torch.float32):
    """(B,C,H,W) fp32 tensor -> NHWC Act of `dtype` (C == 1 fp32 is a free view; CP pads the channel stride with zeros)."""
    _lib.require_gpu(x)
    x = x.contiguous().float()
    B, C, H, W = x.shape
    if C == 1 and CP in (None, 1) and dtype == torch.float32:
        return Act(x.view(B, H, W, 1))
    ld = C if CP is None else CP
    a = Act.empty(B, H, W, C, x.device, ld=ld, zero=ld != C, dtype=dtype)
    _lib.get().call('hv_nchw_to_nhwc', ptr(x), ptr(a.t), a.f16, B, C, H, W, ld, 0, stream())
    return a


# ------------------------------------------------------------------------------------------------ workspace
class _Workspace:
    def __init__(self):
        self.buf = {}
        self.retired = []     # outgrown buffers stay alive: a hipGraph captured earlier still launches kernels that point at them

    def get(self, nbytes, device, slot=0):
        key = (device, slot)
        b = self.buf.get(key)
        if b is None or b.numel() < nbytes:
            if b is not None:
                self.retired.append(b)
            b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.buf[key] = b
        return b


WS = _Workspace()


def _ws(nbytes, device, slot=0):
    # one scratch buffer per (device, slot, stream): kernels on different streams may run concurrently
    b = WS.get(nbytes + 64, device, (slot, torch.cuda.current_stream(device).cuda_stream))
    return b, ctypes.c_size_t(b.numel())


# ------------------------------------------------------------------------------------------------ convolution
def conv_out_size(n, k, stride, pad, dil):
    return (n + 2 * pad - dil * (k - 1) - 1) // stride + 1


def conv2d(x, w, y, k, stride=1, pad=0, dil=1, bias=None, act='none', transposed=False, in_shift=0, alpha=1.0,
           accumulate=0, ch_scale=None, w_bstride=0, ch_scale_bstride=0, precision=None, cin=None, cout=None, w_h=None, mul=None, w_t=None,
           stats=None, _parts_only=False, pool2=False, x1=None, _supported_only=False, wuse=None, bn=None, _bparts_only=False, xn=None):
    """y = act(alpha*ch_scale*conv(x, w) + bias)  [* act'(m) with mul = (Act m, activation name): see hv_conv_desc.mul_src].  x,y: Act; w: prepared [CoutF][k*k][CinP] tensor (or [B][...] with
    w_bstride).  cin/cout default to the view widths (x.C consumed, y.C produced).
    stats: float tensor of conv2d_stats_parts(...) * Cout * 2 elements that receives the per-channel partial sums of the stored output
    (hv_conv_desc.stats; feeds norm_act_forward(partials=...)).
    pool2: y (and mul's tensor) hold the 2x2 sum-pooled output, i.e. half the convolution's own output size (hv_conv_desc.pool2; raises
    RuntimeError 'unsupported' unless the filters-in-LDS kernel serves the shape: pool2_ok()).
    xn = (stats [G][2][Cin], gamma or None, beta or None, groups, activation name, Act out or None): x is the RAW input of a normalisation + activation whose
    statistics norm_act_forward(x, None, ...) left in `stats`; the kernel normalises where it stages x and also stores the normalised map in `out`
    (hv_conv_desc.xn_*; the PatchGAN logits layer only: ask conv2d_supported first)."""
    L = _lib.get()
    d = L.hv_conv_desc()
    kh, kw = (k, k) if isinstance(k, int) else k
    d.x = ptr(x.t).value
    d.B, d.H, d.W = x.B, x.H << in_shift, x.W << in_shift
    d.in_shift, d.x_ld, d.x_coff, d.Cin = in_shift, x.ld, x.coff, (x.C if cin is None else cin)
    d.w = ptr(w).value
    d.w_bstride = w_bstride
    d.Cout, d.KH, d.KW, d.stride, d.pad, d.dil, d.transposed = (y.C if cout is None else cout), kh, kw, stride, pad, dil, int(transposed)
    d.bias = None if bias is None else ptr(bias).value
    d.ch_scale = None if ch_scale is None else ptr(ch_scale).value
    d.ch_scale_bstride = ch_scale_bstride
    d.alpha, d.act, d.accumulate = alpha, ACT[act], int(accumulate)
    d.y = ptr(y.t).value
    d.Ho, d.Wo, d.y_ld, d.y_coff = (y.H << 1 if pool2 else y.H), (y.W << 1 if pool2 else y.W), y.ld, y.coff
    d.pool2 = int(bool(pool2))
    if x1 is not None:      # (Act of ONE channel at the conv's own resolution, full fp32 forward table, its channel index there, row stride, tap stride)
        xa, wfull, ch, w1_row, w1_tap = x1
        assert xa.C == 1 and xa.H == d.H and xa.W == d.W
        d.x1, d.x1_f16, d.x1_ld, d.x1_coff = ptr(xa.t).value, xa.f16, xa.ld, xa.coff
        d.w1, d.w1_row, d.w1_tap = ptr(wfull).value + 4 * ch, w1_row, w1_tap
    d.precision = precision_id(precision)
    d.w_f16 = None if w_h is None else ptr(w_h).value
    d.w_f16_tiled = None if (w_t is None or w_h is None) else ptr(w_t).value      # w_h in MFMA-fragment order (tile_weights / hv_weight_prep)
    d.x_f16, d.y_f16 = x.f16, y.f16
    if mul is not None:
        m, mact = mul
        d.mul_src, d.mul_ld, d.mul_coff, d.mul_act, d.mul_f16 = ptr(m.t).value, m.ld, m.coff, ACT[mact], m.f16
    if bn is not None:      # (Act x_raw, stats [G][2][C], groups, partials or None): batch-norm backward sums out of this data gradient's epilogue (hv_conv_desc.bstats)
        bx, bstat, bgroups, bpart = bn
        assert bx.f16 == y.f16 and bx.H == y.H and bx.W == y.W and bx.B == y.B
        d.bn_x, d.bn_x_ld, d.bn_x_coff = ptr(bx.t).value, bx.ld, bx.coff
        d.bn_stats, d.bn_groups = ptr(bstat).value, int(bgroups)
        if bpart is not None:
            d.bstats = ptr(bpart).value
    if xn is not None:
        nstat, ngam, nbet, ngroups, nact, nout = xn
        d.xn_stats, d.xn_groups, d.xn_act = ptr(nstat).value, int(ngroups), ACT[nact]
        if ngam is not None:
            d.xn_gamma, d.xn_beta = ptr(ngam).value, ptr(nbet).value
        if nout is not None:
            assert nout.f16 and nout.B == x.B and nout.H == x.H and nout.W == x.W
            d.xn_out, d.xn_out_ld, d.xn_out_coff = ptr(nout.t).value, nout.ld, nout.coff
    if _bparts_only:
        return L.size('hv_conv2d_bstats_parts', ctypes.byref(d))
    if _parts_only:
        return L.size('hv_conv2d_stats_parts', ctypes.byref(d))
    if d.Cout == 1:      # single-channel heads / logits: the [pixel][tap] table of conv_head.hip lives in the per-stream scratch
        need = L.size('hv_conv2d_workspace_bytes', ctypes.byref(d))
        if need:
            b, _ = _ws(need, x.t.device, slot=1)
            d.workspace, d.workspace_bytes = ptr(b).value, b.numel()
    if _supported_only:
        return bool(L.cdll.hv_conv2d_supported(ctypes.byref(d)))
    if stats is not None:
        d.stats = ptr(stats).value
    if _TIMER is not None:
        taps = kh * kw if not transposed else max(1, (kh * kw) // (stride * stride))
        flops = 2.0 * y.B * y.H * y.W * d.Cout * taps * d.Cin
        wbytes = (2 if (w_h is not None and d.precision == F16 and not w_bstride) else 4) * d.Cout * kh * kw * d.Cin * (x.B if w_bstride else 1)
        xb, yb = (2 if x.f16 else 4), (2 if y.f16 else 4)      # algorithmic bytes: every operand once, at its storage width
        nbytes = xb * x.B * x.H * x.W * d.Cin + yb * y.B * y.H * y.W * d.Cout * (2 if accumulate else 1) + wbytes + \
            ((2 if mul[0].f16 else 4) * y.B * y.H * y.W * d.Cout if mul is not None else 0)
        _TIMER.wrap(('conv', x.B, d.H, d.W, d.Cin, d.Cout, kh, stride, dil, int(transposed)), flops,
                    lambda: L.call('hv_conv2d', ctypes.byref(d), stream()), nbytes)
    else:
        L.call('hv_conv2d', ctypes.byref(d), stream())
    if wuse is not None:      # (owner, attribute): which of the layer's prepared tables this call's kernel read (engine.ParamSet skips the others when it may)
        setattr(wuse[0], wuse[1], getattr(wuse[0], wuse[1]) | int(L.cdll.hv_last_weight_tables()))
    return y


POOL2 = os.environ.get('HV_POOL2', '1') != '0'      # A/B knob: data gradients of up-sampled inputs written pooled by the conv itself


_SUPPORTED = {}


def conv2d_supported(*args, **kw):
    """Would conv2d(...) with these arguments be served?  (hv_conv2d_supported: the C dispatch itself, run without launching -- the x1 and pool2
    forms have no generic fallback, so their callers ask before they drop the materialised alternative.)"""
    return conv2d(*args, _supported_only=True, **kw)


def pool2_ok(g, y_low, k, stride, pad, dil, precision, w_h, w_t, cout=None, w=None):
    """Can conv2d(..., transposed=True, pool2=True) serve this data gradient?  Asked of the C dispatch (hv_conv2d_supported), once per shape."""
    if not POOL2 or w_h is None or w_t is None or precision_id(precision) != F16 or g.H != 2 * y_low.H or g.W != 2 * y_low.W:
        return False
    co = y_low.C if cout is None else cout
    key = ('pool2', g.B, g.H, g.W, g.C, g.ld, g.coff, g.f16, co, y_low.ld, y_low.coff, y_low.f16, k, stride, pad, dil, ptr(w_t).value & 15)
    r = _SUPPORTED.get(key)
    if r is None:
        r = _SUPPORTED[key] = conv2d_supported(g, w_h if w is None else w, Act(y_low.t, co, y_low.coff), k, stride, pad, dil, transposed=True, pool2=True,
                                               precision=precision, w_h=w_h, w_t=w_t)
    return r


def conv2d_bstats_parts(*args, **kw):
    """Parts of the batch-norm backward sums the conv2d call with these arguments (bn=(x_raw, stats, groups, None)) would write (0: no such epilogue)."""
    return conv2d(*args, _bparts_only=True, **kw)


def conv2d_stats_parts(*args, **kw):
    """Partial-sum rows the conv2d call with these arguments would write into `stats` (0: its kernel has no statistics epilogue)."""
    return conv2d(*args, _parts_only=True, **kw)


class FoldChain:
    """The split-K slab folds of consecutive weight gradients on ONE stream: a call leaves its slabs in one of two alternating scratch buffers and records
    their fold (hv_wgrad_desc.pending); the next call takes the record along (hv_wgrad_desc.carry: the fold runs as extra workgroups of its kernel where
    that kernel has room, else as a launch of its own); flush() launches the last one.  The owner flushes before anything reads the weight gradients."""
    __slots__ = ('pending', 'slot', 'stream')

    def __init__(self, stream_handle):
        self.pending, self.slot, self.stream = None, 0, stream_handle

    def flush(self):
        if self.pending is not None:
            assert torch.cuda.current_stream().cuda_stream == self.stream, 'FoldChain.flush() on another stream than its weight gradients'
            f, self.pending = self.pending, None
            _lib.get().call('hv_wgrad_fold_now', ctypes.byref(f), stream())


FOLD_CHAIN = os.environ.get('HV_FOLD_CHAIN', '1') != '0'      # A/B knob: 0 = every weight gradient folds its slabs in a launch of its own right away


_DIAG_SKIP_WGRAD = frozenset(v for v in os.environ.get('HV_DIAG_SKIP_WGRAD', '').split(',') if v)


def conv2d_wgrad(x, g, dw, k, stride=1, pad=0, dil=1, in_shift=0, accumulate=False, precision=None, cin=None, cout=None, dbias=None,
                 dbias_accumulate=False, chain=None):
    """dw[Cout][k*k][Cin] = sum_pixels g (x) x.  x: conv input view, g: gradient wrt the conv output.
    dbias: optional [Cout] tensor that receives sum_pixels g (the bias gradient), computed by the same kernels.
    chain: a FoldChain of the current stream -- this call's slab fold is recorded on it (and done inside the next call of the chain or by chain.flush())
    and the chain's previous fold is carried along."""
    L = _lib.get()
    d = L.hv_wgrad_desc()
    kh, kw = (k, k) if isinstance(k, int) else k
    if _DIAG_SKIP_WGRAD:      # timing-only diagnostic (wrong parameter gradients): what a class of weight gradients costs the step
        ci_, co_ = (x.C if cin is None else cin), (g.C if cout is None else cout)
        cls = {'k%d' % kh, 'thin' if min(ci_, co_) <= 16 else 'wide', 'head' if co_ <= 4 else '', 'dstem' if (kh == 4 and stride == 2 and ci_ <= 4) else '',
               'g3thin' if (kh == 3 and min(ci_, co_) <= 16 and co_ > 4) else '', 'dhead' if (kh == 4 and co_ <= 4) else '', 'ghead' if (kh == 3 and co_ <= 4) else ''}
        if cls & _DIAG_SKIP_WGRAD:
            return
    d.x = ptr(x.t).value
    d.B, d.H, d.W, d.in_shift = x.B, x.H << in_shift, x.W << in_shift, in_shift
    d.x_ld, d.x_coff, d.Cin = x.ld, x.coff, (x.C if cin is None else cin)
    d.g = ptr(g.t).value
    d.Ho, d.Wo, d.g_ld, d.g_coff, d.Cout = g.H, g.W, g.ld, g.coff, (g.C if cout is None else cout)
    d.KH, d.KW, d.stride, d.pad, d.dil = kh, kw, stride, pad, dil
    d.dw = ptr(dw).value
    d.accumulate = int(accumulate)
    d.precision = precision_id(precision)
    d.dbias = None if dbias is None else ptr(dbias).value
    d.dbias_accumulate = int(dbias_accumulate)
    d.x_f16, d.g_f16 = x.f16, g.f16
    d.workspace, d.workspace_bytes = None, 0
    if chain is not None and not FOLD_CHAIN:
        chain = None
    need = L.size('hv_conv2d_wgrad_workspace_bytes', ctypes.byref(d))
    fold = None
    if chain is not None:
        assert torch.cuda.current_stream(x.t.device).cuda_stream == chain.stream, 'a FoldChain belongs to one stream'
        if chain.pending is not None:
            d.carry = ctypes.pointer(chain.pending)
        if need:
            b, _ = _ws(need, x.t.device, slot=('fold-chain', chain.slot))      # (the carried fold reads the OTHER buffer)
            chain.slot ^= 1
            d.workspace, d.workspace_bytes = ptr(b).value, b.numel()
            fold = L.hv_wgrad_fold()
            d.pending = ctypes.pointer(fold)
    elif need:
        b, _ = _ws(need, x.t.device)
        d.workspace, d.workspace_bytes = ptr(b).value, b.numel()
    if _TIMER is not None:
        flops = 2.0 * g.B * g.H * g.W * d.Cout * kh * kw * d.Cin
        nbytes = (2 if x.f16 else 4) * x.B * x.H * x.W * d.Cin + (2 if g.f16 else 4) * g.B * g.H * g.W * d.Cout + 4 * d.Cout * kh * kw * d.Cin   # x and g read once, dW written once
        _TIMER.wrap(('wgrad', x.B, d.H, d.W, d.Cin, d.Cout, kh, stride, dil, 0), flops,
                    lambda: L.call('hv_conv2d_wgrad', ctypes.byref(d), stream()), nbytes)
    else:
        L.call('hv_conv2d_wgrad', ctypes.byref(d), stream())
    if chain is not None:      # the carried fold is done (inside this call's kernel or beside it); this call's own is the chain's pending one now
        chain.pending = fold if (fold is not None and fold.nslabs > 0) else None
    return dw


# ------------------------------------------------------------------------------------------------ weight prep tables
class LayerTable:
    """Device array of per-layer descriptors for the batched weight-prep kernels.  Rebuilt when a
    parameter's storage moves (e.g. after .to(device) / load_state_dict on a fresh module)."""

    def __init__(self, struct_name):
        self.struct_name = struct_name
        self.key = None
        self.dev = None
        self.n = 0

    def update(self, rows, key, device):
        """rows: list of dicts field -> value (tensors are converted to pointers)."""
        if key == self.key:
            return
        L = _lib.get()
        S = getattr(L, self.struct_name)
        arr = (S * len(rows))()
        for i, r in enumerate(rows):
            for f, v in r.items():
                if torch.is_tensor(v):
                    v = ptr(v).value
                setattr(arr[i], f, v)
        raw = bytes(arr)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self.dev = host.to(device)
        self.key, self.n = key, len(rows)

    def ptr(self):
        return ptr(self.dev)


def tiled_elems(rows, taps, K):
    """halfs of the MFMA-fragment-ordered copy of an fp16 filter table [rows][taps][K] (0: this shape has none)."""
    return _lib.get().size('hv_weight_tiled_elems', int(rows), int(taps), int(K))


def tile_weights(w_h, rows, taps, K):
    """fp16 filter table [rows][taps][K] -> its MFMA-fragment-ordered copy (hv_conv_desc.w_f16_tiled), or None when the shape has none."""
    n = tiled_elems(rows, taps, K)
    if not n:
        return None
    out = torch.empty(n, dtype=torch.float16, device=w_h.device)
    _lib.get().call('hv_weight_tile_f16', ptr(w_h), ptr(out), int(rows), int(taps), int(K), stream())
    return out


def weight_prep(table, max_numel, any_sn=True, any_legacy=True):
    _lib.get().call('hv_weight_prep2', ctypes.cast(table.ptr(), ctypes.POINTER(_lib.get().hv_wprep_layer)), table.n,
                    ctypes.c_longlong(max_numel), int(any_sn), int(any_legacy), stream())


def weight_prep_backward(table, max_numel, any_sn):
    _lib.get().call('hv_weight_prep_backward', ctypes.cast(table.ptr(), ctypes.POINTER(_lib.get().hv_wprep_bwd_layer)), table.n,
                    ctypes.c_longlong(max_numel), int(any_sn), stream())


# ------------------------------------------------------------------------------------------------ pointwise
_DIAG_SKIP = os.environ.get('HV_DIAG_SKIP', '')      # timing-only diagnostics (tools/marginal_step.sh): the named passes are not launched, results are wrong


def act_backward(dy, y, act, dbias=None, dbias_accumulate=False):
    """In place: dy *= act'(y); dbias (+)= column sums."""
    if 'act_bwd' in _DIAG_SKIP:
        return
    L = _lib.get()
    need = L.size('hv_act_backward_workspace_bytes', ctypes.c_longlong(dy.npix), dy.C) if dbias is not None else 0
    b, nb = _ws(need, dy.t.device)
    L.call('hv_act_backward', ptr(dy.t), dy.f16, ptr(y.t), y.f16, ctypes.c_longlong(dy.npix), dy.C, dy.ld, dy.coff, y.ld, y.coff, ACT[act],
           ptr(dbias), int(dbias_accumulate), ptr(b), nb, stream())


def head_seed_backward(seed, y, carrier, act, dbias=None, dbias_accumulate=False):
    """1-channel head: carrier[..., 0] = seed * act'(y) (fp16 [B,H,W,4] carrier, channels 1-3 zeroed), dbias (+)= its sum.  seed: fp32 tensor of npix
    elements; y: Act (the head's output)."""
    L = _lib.get()
    npix = carrier.npix
    need = L.size('hv_head_seed_workspace_bytes', ctypes.c_longlong(npix)) if dbias is not None else 0
    b, nb = _ws(need, carrier.t.device)
    L.call('hv_head_seed_backward', ptr(seed), ptr(y.t), y.f16, y.ld, y.coff, ptr(carrier.t), ctypes.c_longlong(npix), ACT[act], ptr(dbias),
           int(dbias_accumulate), ptr(b), nb, stream())


def copy_channels(src, dst, mode=0, accumulate=False):
    """dst (+)= resample(src); H,W of dst rule (mode: 0 same, 1 up x2, 2 down x1/2, 3 adjoint of 1, 4 adjoint of 2)."""
    if 'copy_channels' in _DIAG_SKIP:
        return
    assert src.C == dst.C
    _lib.get().call('hv_copy_channels', ptr(src.t), src.f16, ptr(dst.t), dst.f16, dst.B, dst.H, dst.W, dst.C, src.ld, src.coff, dst.ld, dst.coff,
                    mode, int(accumulate), stream())


def add_channels(a, b, dst):
    """dst = a + b (Acts of the same size and channel count, any storage): one launch for a gradient with two contributions."""
    assert a.C == b.C == dst.C and a.npix == b.npix == dst.npix
    _lib.get().call('hv_add_channels', ptr(a.t), a.f16, a.ld, a.coff, ptr(b.t), b.f16, b.ld, b.coff, ptr(dst.t), dst.f16, dst.ld, dst.coff,
                    ctypes.c_longlong(dst.npix), dst.C, stream())


def fill(t, value=0.0):
    _lib.get().call('hv_fill', ptr(t), ctypes.c_longlong(t.numel()), ctypes.c_float(value), stream())


def axpy(y, x, a=1.0):
    assert y.numel() == x.numel()
    _lib.get().call('hv_axpy', ptr(y), ptr(x), ctypes.c_longlong(y.numel()), ctypes.c_float(a), stream())


def gen_input(x, seg, mask, ratio, dst, order):
    _lib.get().call('hv_gen_input', ptr(x), ptr(seg), ptr(mask), ptr(ratio), ptr(dst.t), dst.f16, dst.B, dst.H, dst.W, dst.ld, order, stream())


def gap_fc_sigmoid(x, fc_w, fc_b, pooled, pred):
    L = _lib.get()
    need = L.size('hv_gap_fc_workspace_bytes', x.B, x.C)
    b, nb = _ws(need, x.t.device)
    L.call('hv_gap_fc_sigmoid', ptr(x.t), x.f16, x.B, x.H * x.W, x.C, x.ld, ptr(fc_w), ptr(fc_b), ptr(pooled), ptr(pred), ptr(b), nb, stream())


def gap_fc_sigmoid_backward(dpred, pred, pooled, fc_w, dx, dw, db, accumulate=False, mul=None):
    """mul = (Act m, activation name): the contribution to dx is multiplied by act'(m) (dx holds pre-activation gradients, see conv2d's mul)."""
    m, mact = mul if mul is not None else (None, 'none')
    _lib.get().call('hv_gap_fc_sigmoid_backward', ptr(dpred), ptr(pred), ptr(pooled), ptr(fc_w), ptr(dx.t), dx.f16, dx.B, dx.H * dx.W, dx.C,
                    dx.ld, ptr(dw), ptr(db), int(accumulate), None if m is None else ptr(m.t), 0 if m is None else m.f16, 0 if m is None else m.ld,
                    ACT[mact], stream())


def sobel(img, out=None):
    """img: (B,1,H,W) tensor."""
    out = torch.empty_like(img) if out is None else out
    B, _, H, W = img.shape
    _lib.get().call('hv_sobel', ptr(img), ptr(out), B, H, W, stream())
    return out


GAN_LOSS_WS = os.environ.get('HV_GAN_LOSS_WS', '1') != '0'   # A/B knob


def gan_loss(z, target_is_real, mode='vanilla', loss=None, loss_weight=1.0, loss_accumulate=False, dz=None, grad_weight=1.0, carrier=None, dbias=None,
             dbias_accumulate=False):
    """carrier: Act fp16 [.., 4] -- the logits layer's padded gradient operand, written directly (hv_gan_loss_head), with the layer's bias gradient dbias."""
    m = {'vanilla': 0, 'lsgan': 1}[mode]
    L = _lib.get()
    n = z.numel()
    if carrier is not None:
        assert carrier.f16 and carrier.ld == 4 and carrier.coff == 0 and carrier.npix == n
        b, nb = _ws(L.size('hv_gan_loss_head_workspace_bytes', ctypes.c_longlong(n)), z.device, slot=2)
        L.call('hv_gan_loss_head', ptr(z), ctypes.c_longlong(n), int(bool(target_is_real)), m, ctypes.c_float(loss_weight), ptr(loss), int(loss_accumulate),
               ctypes.c_float(grad_weight), ptr(dz), ptr(carrier.t), ptr(dbias), int(dbias_accumulate), ptr(b), nb, stream())
        return
    if n >= 4096 and GAN_LOSS_WS:      # many workgroups + per-stream scratch for their partial sums
        b, nb = _ws(L.size('hv_gan_loss_workspace_bytes', ctypes.c_longlong(n)), z.device, slot=2)
        L.call('hv_gan_loss_ws', ptr(z), ctypes.c_longlong(n), int(bool(target_is_real)), m, ctypes.c_float(loss_weight), ptr(loss),
               int(loss_accumulate), ctypes.c_float(grad_weight), ptr(dz), ptr(b), nb, stream())
        return
    L.call('hv_gan_loss', ptr(z), ctypes.c_longlong(n), int(bool(target_is_real)), m, ctypes.c_float(loss_weight),
           ptr(loss), int(loss_accumulate), ctypes.c_float(grad_weight), ptr(dz), stream())


GAN_LOSS_PAIR = os.environ.get('HV_GAN_LOSS_PAIR', '1') != '0'   # A/B knob


_TICKETS = {}


def _ticket(device):
    """One zero-initialised counter per (device, stream) for the kernels whose last workgroup folds their partial results (they leave it at zero)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream)
    t = _TICKETS.get(key)
    if t is None:
        t = _TICKETS[key] = torch.zeros(16, dtype=torch.int32, device=device)
    return t


def gan_loss_pair(z0, real0, loss0, carrier0, z1=None, real1=True, loss1=None, carrier1=None, mode='vanilla', loss_weight=1.0, loss_accumulate=False,
                  grad_weight=1.0, dbias=None, dbias_accumulate=False):
    """The PatchGAN loss head of one or two logit ranges (the fake | real halves of a batched discriminator pass) in ONE launch (hv_gan_loss_head_pair: the
    last workgroup folds the block sums).  Returns False when switched off (the caller then takes gan_loss per range)."""
    if not GAN_LOSS_PAIR:
        return False
    n0, n1 = z0.numel(), (0 if z1 is None else z1.numel())
    for c, n in ((carrier0, n0), (carrier1, n1)):
        assert c is None or (c.f16 and c.ld == 4 and c.coff == 0 and c.npix == n)
    L = _lib.get()
    b, nb = _ws(L.size('hv_gan_loss_head_pair_workspace_bytes', ctypes.c_longlong(n0), ctypes.c_longlong(n1)), z0.device, slot=2)
    L.call('hv_gan_loss_head_pair', ptr(z0), ctypes.c_longlong(n0), int(bool(real0)), ptr(loss0), ptr(carrier0.t), ptr(z1), ctypes.c_longlong(n1),
           int(bool(real1)), ptr(loss1), None if carrier1 is None else ptr(carrier1.t), {'vanilla': 0, 'lsgan': 1}[mode], ctypes.c_float(loss_weight),
           int(loss_accumulate), ctypes.c_float(grad_weight), ptr(dbias), int(dbias_accumulate), ptr(b), nb,
           ptr(_ticket(z0.device)), stream())
    return True


def adam_step(table, max_numel, lr_dev, beta1, beta2, eps, step_dev, guard_flat=None, grad_mul=1.0):
    L = _lib.get()
    if guard_flat is not None:      # step_dev: 8 floats (hv_adam_step_guarded)
        L.call('hv_adam_step_guarded', ctypes.cast(table.ptr(), ctypes.POINTER(L.hv_adam_tensor)), table.n, ctypes.c_longlong(max_numel),
               ptr(lr_dev), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ptr(step_dev), ptr(guard_flat),
               ctypes.c_longlong(guard_flat.numel()), ctypes.c_float(grad_mul), stream())
        return
    L.call('hv_adam_step', ctypes.cast(table.ptr(), ctypes.POINTER(L.hv_adam_tensor)), table.n, ctypes.c_longlong(max_numel),
           ptr(lr_dev), ctypes.c_float(beta1), ctypes.c_float(beta2), ctypes.c_float(eps), ptr(step_dev), stream())


# ------------------------------------------------------------------------------------------------ norm + activation
def norm_act_forward(x, y, norm, training, stats, gamma=None, beta=None, running_mean=None, running_var=None, nbt=None,
                     act='lrelu', post_sigmoid=False, eps=1e-5, momentum=0.1, groups=1, partials=None, n_partials=0):
    L = _lib.get()
    d = L.hv_norm_desc()
    d.x = ptr(x.t).value
    d.B, d.HW, d.C = x.B, x.H * x.W, x.C
    d.x_ld, d.x_coff = x.ld, x.coff
    if y is not None:      # (None: statistics only -- the consumer normalises at its own staging: conv2d(xn=...))
        d.y, d.y_ld, d.y_coff = ptr(y.t).value, y.ld, y.coff
    d.norm, d.training, d.eps, d.momentum = NORM[norm], int(training), eps, momentum
    for f, v in (('gamma', gamma), ('beta', beta), ('running_mean', running_mean), ('running_var', running_var),
                 ('num_batches_tracked', nbt), ('stats', stats)):
        setattr(d, f, None if v is None else ptr(v).value)
    d.act, d.post_sigmoid, d.groups = ACT[act], int(post_sigmoid), int(groups)
    assert y is None or x.f16 == y.f16
    d.f16 = x.f16
    if partials is not None and n_partials:      # the producing conv's own statistics (hv_conv_desc.stats): no reduction pass over x
        d.partials, d.n_partials = ptr(partials).value, int(n_partials)
    need = L.size('hv_norm_workspace_bytes', x.B, x.H * x.W, x.C)
    b, _ = _ws(need, x.t.device)
    d.workspace, d.workspace_bytes = ptr(b).value, b.numel()
    L.call('hv_norm_act_forward', ctypes.byref(d), stream())


def norm_act_backward(dy, y, x, dx, norm, training, stats, gamma=None, act='lrelu', post_sigmoid=False, dgamma=None, dbeta=None,
                      param_accumulate=False, groups=1, partials=None, n_partials=0):
    L = _lib.get()
    d = L.hv_norm_bwd_desc()
    d.dy, d.y, d.x, d.dx = ptr(dy.t).value, ptr(y.t).value, ptr(x.t).value, ptr(dx.t).value
    d.B, d.HW, d.C = x.B, x.H * x.W, x.C
    d.dy_ld, d.dy_coff, d.y_ld, d.y_coff = dy.ld, dy.coff, y.ld, y.coff
    d.x_ld, d.x_coff, d.dx_ld, d.dx_coff = x.ld, x.coff, dx.ld, dx.coff
    d.norm, d.training = NORM[norm], int(training)
    d.gamma = None if gamma is None else ptr(gamma).value
    d.stats = ptr(stats).value
    d.act, d.post_sigmoid = ACT[act], int(post_sigmoid)
    d.dgamma = None if dgamma is None else ptr(dgamma).value
    d.dbeta = None if dbeta is None else ptr(dbeta).value
    d.param_accumulate = int(param_accumulate)
    d.groups = int(groups)
    assert dy.f16 == y.f16 == x.f16 == dx.f16
    d.f16 = x.f16
    if partials is not None and n_partials:      # the sums came out of the data gradient that wrote dy (hv_conv_desc.bstats): no reduction pass here
        d.partials, d.n_partials = ptr(partials).value, int(n_partials)
    need = L.size('hv_norm_workspace_bytes', x.B, x.H * x.W, x.C)
    b, _ = _ws(need, x.t.device)
    d.workspace, d.workspace_bytes = ptr(b).value, b.numel()
    L.call('hv_norm_act_backward', ctypes.byref(d), stream())
