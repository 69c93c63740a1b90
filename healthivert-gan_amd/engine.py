"""Explicit forward/backward executors of the hot path: sequences of libhvgan kernel launches over
pre-allocated NHWC buffers, with no autograd tape, no host synchronisation and no allocation after
the first call for a given shape (so a whole train step can be captured in a hipGraph).

The nn.Modules of `models/` hold the parameters (reference state-dict keys) and delegate here.
"""
import contextlib
import ctypes
import os

import torch

from . import lib as _lib
from . import ops
from .lib import ptr, stream
from .ops import Act, rup


def precision_note(precision):
    """What the precision mode means, for bench.py's config record."""
    if precision in ('fp16', 'f16'):
        return 'fp16 MFMA operands / fp32 accumulate; activations and gradients stored as fp16 (image tensors, attention scores, weights, statistics fp32)'
    return 'fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 storage'


# ================================================================================================ parameters
class ConvParams:
    """One convolution's parameters and their kernel-layout copies.

    weight: [cout,cin,k,k] (or [cin,cout,k,k] with transposed_src, i.e. nn.ConvTranspose2d).
    cin_fwd / cin_wg: channel width consumed by the forward gather / by the weight-gradient kernel
    (equal except for a 1-channel image input, which the forward reads unpadded)."""

    def __init__(self, name, weight, bias, cin, cout, k, cin_fwd=None, cin_wg=None, u=None, v=None, transposed_src=False):
        self.name, self.weight, self.bias = name, weight, bias
        self.cin, self.cout, self.k, self.taps = cin, cout, k, k * k
        self.cin_fwd = rup(cin, 4) if cin_fwd is None else cin_fwd
        self.cin_wg = rup(cin, 4) if cin_wg is None else cin_wg
        self.coutP = rup(cout, 4)
        self.u, self.v = u, v
        self.sn = u is not None
        self.transposed_src = transposed_src
        self.w_fwd = self.w_bwd = self.sigma = self.dw = None
        self.w_fwd_t = self.w_bwd_t = None      # fp16 tables in MFMA-fragment order (None: the shape has no tiled form)
        # which prepared tables the layer's forward / data-gradient convolutions have read so far (hv_last_weight_tables bits: 1 fp32, 2 fp16 rows, 4 fp16
        # fragment order; 0 = not seen yet): ParamSet.prep writes only those inside a lean_tables() context
        self.use_fwd = self.use_bwd = 0
        self.owner = None
        self.split_k = None                     # K2: also keep the first K2 input channels as a fragment-ordered table of their own (w_fwd_t2: conv2d's x1 layers)
        self.w_fwd_t2 = None

    def sizes(self):
        return (self.cout * self.taps * self.cin_fwd, self.cin_fwd * self.taps * self.coutP, self.coutP * self.taps * self.cin_wg)


class ParamSet:
    """All convolutions of one network: batched weight preparation (spectral norm + layouts), batched
    weight-gradient finalisation, flat gradient storage (one all-reduce per network under DDP)."""

    def __init__(self, convs, extra_params=()):
        self.convs = list(convs)
        self.extra = list(extra_params)          # other trainable tensors (fc, BN affine, biases are added automatically)
        self.device = None
        self.t_prep = {True: ops.LayerTable('hv_wprep_layer'), False: ops.LayerTable('hv_wprep_layer')}
        self.t_bwd = {True: ops.LayerTable('hv_wprep_bwd_layer'), False: ops.LayerTable('hv_wprep_bwd_layer')}
        # lean tables by (parameter storage, power iteration, the layers' table-use bits): a captured step graph keeps the DEVICE address of the table it was
        # captured with, and the use bits can still grow afterwards (another batch shape's first eager step, an eval forward at another size), so a table
        # is never rebuilt in place or freed -- a new bit pattern gets a new table and the old graphs keep theirs (a superset of bits stays correct for them:
        # the layout pass then merely writes a table they do not read)
        self.t_lean = {}
        # the slab-fold chains of this network's weight gradients, one per stream they are issued on (ops.FoldChain)
        self.fold_chains = {}
        for c in self.convs:
            c.owner = self
        self.flat_grad = None
        self._key = None
        # which weights the prepared tables belong to: `version` counts the changes of the parameters that this code knows of (optimiser steps, state-dict
        # loads, new storage), `prepared` = (version, table set) of the last prep launch.  prep(only_if_stale=True) skips the launch when both still match --
        # the discriminators' tables written for the generator's part of step t are the ones the discriminator update of step t + 1 reads
        self.version = 0
        self.prepared = None

    def trainable(self):
        seen, out = set(), []
        for c in self.convs:
            for p in (c.weight, c.bias):
                if p is not None and id(p) not in seen:
                    seen.add(id(p))
                    out.append(p)
        for p in self.extra:
            if id(p) not in seen:
                seen.add(id(p))
                out.append(p)
        return out

    def _ensure(self, device):
        key = tuple(p.data_ptr() for p in self.trainable()) + tuple(c.u.data_ptr() for c in self.convs if c.sn)
        if key == self._key and self.device == device:
            return
        self._key, self.device = key, device
        self.version += 1
        tot = sum(sum(c.sizes()) + 4 for c in self.convs)
        store = torch.zeros(tot, dtype=torch.float32, device=device)
        off = 0
        for c in self.convs:
            a, b, d = c.sizes()
            c.w_fwd = store[off:off + a]; off += a
            c.w_bwd = store[off:off + b]; off += b
            c.dw = store[off:off + d]; off += d
            c.sigma = store[off:off + 4]; off += 4
            c.sigma[0] = 1.0            # (layers without spectral norm: hv_weight_prep2 skips the sigma kernel when a net has none)
        self._store = store
        # fp16 copies of the prepared weights (halo-tiled conv kernel, HV_F16 precision)
        hstore = torch.zeros(sum(c.sizes()[0] + c.sizes()[1] + 16 for c in self.convs), dtype=torch.float16, device=device)
        off = 0
        for c in self.convs:
            a, b, _ = c.sizes()
            c.w_fwd_h = hstore[off:off + a]; off += (a + 7) // 8 * 8
            c.w_bwd_h = hstore[off:off + b]; off += (b + 7) // 8 * 8
        self._hstore = hstore
        # the same fp16 tables in MFMA-fragment order, for the kernels that fetch filter rows straight into MFMA registers (zero-filled once:
        # the prep kernel never writes the padding rows)
        tsz = [(ops.tiled_elems(c.cout, c.taps, c.cin_fwd), ops.tiled_elems(c.cin_fwd, c.taps, c.coutP),
                ops.tiled_elems(c.cout, c.taps, c.split_k) if c.split_k else 0) for c in self.convs]
        tstore = torch.zeros(sum(a + b + e for a, b, e in tsz) + 8, dtype=torch.float16, device=device)
        off = 0
        for c, (a, b, e) in zip(self.convs, tsz):
            c.w_fwd_t = tstore[off:off + a] if a else None; off += a
            c.w_bwd_t = tstore[off:off + b] if b else None; off += b
            c.w_fwd_t2 = tstore[off:off + e] if e else None; off += e
        self._tstore = tstore
        # flat gradients; .grad of every trainable tensor is a view into it
        ps = self.trainable()
        n = sum(p.numel() for p in ps)
        # (grad_home: a caller-provided slice of a larger buffer -- the data-parallel step keeps the three discriminators' flat gradients
        # back to back so that ONE all-reduce averages them all)
        home = getattr(self, 'grad_home', None)
        if home is not None and home.numel() == n and home.device == torch.device(device) and home.dtype == torch.float32:
            self.flat_grad = home
            self.flat_grad.zero_()
        else:
            self.flat_grad = torch.zeros(n, dtype=torch.float32, device=device)
        off = 0
        for p in ps:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        for pi in (True, False):
            self.t_prep[pi].update(self._prep_rows(pi, False), key, device)
        for acc in (True, False):
            rows = []
            for c in self.convs:
                rows.append(dict(dw_ohwi=c.dw, w_fwd=c.w_fwd, u=c.u if c.sn else None, v=c.v if c.sn else None, sigma=c.sigma,
                                 dw_orig=c.weight.grad, Cout=c.cout, Cin=c.cin, taps=c.taps,
                                 CinP=c.coutP if c.transposed_src else c.cin_wg, sn=int(c.sn),
                                 transposed_src=int(c.transposed_src), accumulate=int(acc)))
            self.t_bwd[acc].update(rows, key, device)

    def attach_grads(self):
        """Re-point .grad at the flat buffer (optimizer.zero_grad(set_to_none=True) drops the views)."""
        off = 0
        for p in self.trainable():
            n = p.numel()
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + n].view_as(p)
            off += n

    def _prep_rows(self, pi, lean):
        rows = []
        for c in self.convs:
            # lean: only the tables the layer's convolutions have been seen to read (0 = not seen yet: all of them); conv_transpose sources keep all (their
            # layout kernels write every table), spectral-norm layers keep the fp32 forward table (hv_weight_prep_backward reads it)
            mf = mb = 7
            if lean and not c.transposed_src:
                mf = (c.use_fwd or 7) | (1 if c.sn else 0)
                mb = c.use_bwd or 7
            rows.append(dict(w_orig=c.weight.data, u=c.u if c.sn else None, v=c.v if c.sn else None, sigma=c.sigma,
                             w_fwd=c.w_fwd if mf & 1 else None, w_bwd=c.w_bwd if mb & 1 else None, w_fwd_h=c.w_fwd_h if mf & 2 else None,
                             w_bwd_h=c.w_bwd_h if mb & 2 else None, w_fwd_t=c.w_fwd_t if mf & 4 else None, w_bwd_t=c.w_bwd_t if mb & 4 else None,
                             Cout=c.cout, Cin=c.cin, taps=c.taps, CinP=c.cin_fwd, CoutF=c.cout, CoutP=c.coutP, CinB=c.cin_fwd, sn=int(c.sn),
                             power_iter=int(pi and c.sn), transposed_src=int(c.transposed_src), w_fwd_t2=c.w_fwd_t2, K2=int(c.split_k or 0)))
        return rows

    def weights_changed(self):
        """The parameters were written by something other than this ParamSet's own launches (an optimiser step, load_state_dict, user code): the next
        prep(only_if_stale=True) must run."""
        self.version += 1

    def prep(self, device, power_iter, only_if_stale=False):
        self._ensure(device)
        pi = bool(power_iter)
        table = self.t_prep[pi]
        tset = 'full'
        if LEAN and LEAN_TABLES and ops.precision_id(None) == ops.F16:
            tset = tuple((c.use_fwd, c.use_bwd) for c in self.convs)
        if only_if_stale and not pi and PREP_SKIP and self.prepared is not None and self.prepared[0] == self.version and self.prepared[1] in ('full', tset):
            return      # the tables in memory were written from these very weights (and cover the tables asked for)
        self.prepared = (self.version, tset)
        if LEAN and LEAN_TABLES and ops.precision_id(None) == ops.F16:
            # inside the train step, after the step has been seen once for this batch shape: the layout pass skips the tables no kernel of the layer reads
            # (fp16 mode: the big layers read the fragment-ordered tables only -- 4 of 20 bytes per weight)
            key = (self._key, pi, tuple((c.use_fwd, c.use_bwd) for c in self.convs))
            table = self.t_lean.get(key)
            if table is None:
                table = self.t_lean[key] = ops.LayerTable('hv_wprep_layer')
                table.update(self._prep_rows(pi, True), key, device)
        ops.weight_prep(table, max(c.sizes()[0] + c.sizes()[1] for c in self.convs), any_sn=any(c.sn for c in self.convs),
                        any_legacy=any(c.transposed_src for c in self.convs))

    def fold_chain(self):
        """The FoldChain of the current stream (weight gradients of this network issued there carry each other's slab folds)."""
        h = torch.cuda.current_stream().cuda_stream
        c = self.fold_chains.get(h)
        if c is None:
            c = self.fold_chains[h] = ops.FoldChain(h)
        return c

    def flush_folds(self):
        """The current stream's last recorded fold as a launch of its own; chains of other streams must have been flushed there before the join."""
        h = torch.cuda.current_stream().cuda_stream
        for k, c in self.fold_chains.items():
            if k == h:
                c.flush()
            else:
                assert c.pending is None, 'a weight-gradient fold chain of another stream was not flushed before finish_backward()'

    def finish_backward(self, accumulate=False):
        """Kernel-layout weight gradients -> .grad of weight_orig / weight (spectral-norm backward included)."""
        self.flush_folds()
        ops.weight_prep_backward(self.t_bwd[bool(accumulate)], max(c.cout * c.cin * c.taps for c in self.convs), any(c.sn for c in self.convs))


class ConvNode:
    """conv (+bias +activation) between two NHWC views; knows how to run forward and backward."""
    __slots__ = ('p', 'x', 'y', 'k', 's', 'pad', 'd', 'act', 'shift', 'need_dx', 'transposed', 'use_bias', 'dx_c', 'pool_to', 'split', '_split_ok')

    def __init__(self, p, x, y, s=1, pad=0, d=1, act='none', shift=0, need_dx=True, transposed=False, use_bias=True, dx_c=None):
        self.p, self.x, self.y, self.k, self.s, self.pad, self.d = p, x, y, p.k, s, pad, d
        self.act, self.shift, self.need_dx, self.transposed, self.use_bias = act, shift, need_dx, transposed, use_bias
        # dx_c: only the first dx_c input channels' gradient is wanted (a concat input whose tail channels are network inputs): the data
        # gradient is then a convolution with fewer output channels (the first dx_c rows of the transposed filter table)
        self.dx_c = dx_c
        # pool_to = (Act low, activation of low's producer): x is a concat buffer whose first dx_c channels are the nearest x2 up-sampling of `low`;
        # set by the owner of the plan when the pooled data gradient can serve it (conv_backward)
        self.pool_to = None
        # split = (Act low, Act x1): x is the concat [nearest x2 up-sampling of low (p.split_k channels) | x1 (one channel) | padding]; where the
        # filters-in-LDS kernel serves the shape the forward reads `low` with the fused up-sampling and adds x1's taps in its epilogue, so the
        # up-sampled part of x is only materialised for the backward (split_forward() tells)
        self.split = None
        self._split_ok = None

    def split_forward(self, prec):
        """True when the forward reads [up-sampled low | x1] without the concat: asked of the C dispatch (hv_conv2d_supported -- the x1 kernel exists
        for two channel shapes only and follows the HV_CONV_LF knobs), once per node; otherwise the materialised concat is built and read."""
        p = self.p
        if self.split is None or not SPLIT_CONCAT or p.w_fwd_t2 is None or ops.precision_id(prec) != ops.F16:
            return False
        key = (ops.precision_id(prec), p.w_fwd_t2.data_ptr(), self.y.t.data_ptr())
        if getattr(self, '_split_ok', None) is None or self._split_ok[0] != key:
            low, x1 = self.split
            ok = bool(low.f16 and self.y.f16 and low.C == p.split_k) and ops.conv2d_supported(
                low, p.w_fwd, self.y, self.k, self.s, self.pad, self.d, bias=p.bias if self.use_bias else None, act=self.act, w_h=p.w_fwd_h, w_t=p.w_fwd_t2,
                in_shift=1, precision=prec, cin=p.split_k, cout=p.cout, x1=(x1, p.w_fwd, p.split_k, p.taps * p.cin_fwd, p.cin_fwd))
            self._split_ok = (key, ok)
        return self._split_ok[1]

    def forward(self, prec, stats=None, xn=None, probe=False, x_raw=None):
        """xn: the input is normalised + activated where the kernel stages it (ops.conv2d) -- x_raw is then the normalisation's raw input, read instead
        of self.x (which xn's `out` fills); probe=True only asks whether that form is served."""
        p = self.p
        if self.split_forward(prec):
            low, x1 = self.split
            ops.conv2d(low, p.w_fwd, self.y, self.k, self.s, self.pad, self.d, bias=p.bias if self.use_bias else None, act=self.act, w_h=p.w_fwd_h,
                       w_t=p.w_fwd_t2, in_shift=1, precision=prec, cin=p.split_k, cout=p.cout,
                       x1=(x1, p.w_fwd, p.split_k, p.taps * p.cin_fwd, p.cin_fwd), wuse=(p, 'use_fwd'))
            return
        src = self.x if x_raw is None else x_raw
        xin = Act(src.t, p.cin_fwd, src.coff)
        if probe:
            return ops.conv2d_supported(xin, p.w_fwd, self.y, self.k, self.s, self.pad, self.d, bias=p.bias if self.use_bias else None, act=self.act, w_h=p.w_fwd_h,
                                        w_t=p.w_fwd_t, in_shift=self.shift, transposed=self.transposed, precision=prec, cout=p.cout, xn=xn)
        ops.conv2d(xin, p.w_fwd, self.y, self.k, self.s, self.pad, self.d, bias=p.bias if self.use_bias else None, act=self.act, w_h=p.w_fwd_h, w_t=p.w_fwd_t,
                   in_shift=self.shift, transposed=self.transposed, precision=prec, cout=p.cout, stats=stats, wuse=(p, 'use_fwd'), xn=xn)

    def stats_parts(self, prec):
        """Partial-sum rows this node's forward kernel writes when handed a statistics buffer (0: that kernel has no such epilogue)."""
        p = self.p
        xin = Act(self.x.t, p.cin_fwd, self.x.coff)
        return ops.conv2d_stats_parts(xin, p.w_fwd, self.y, self.k, self.s, self.pad, self.d, bias=p.bias if self.use_bias else None, act=self.act,
                                      w_h=p.w_fwd_h, w_t=p.w_fwd_t, in_shift=self.shift, transposed=self.transposed, precision=prec, cout=p.cout)


# streams on which weight gradients stay in line instead of forking to a side stream: the per-discriminator streams of
# the train step (three chains already run concurrently there, and a fork nested inside a forked stream crashes
# hipStreamEndCapture on ROCm 7.2 when the step is captured as a graph)
NO_FORK_STREAMS = set()
_named_streams = {}


def named_stream(name, device, priority=0):
    """Process-wide HIP stream for a role (a discriminator's stream, the capture stream, the weight-gradient side stream of a parent stream ...).
    torch hands out streams from a pool of 32 per device and priority, round-robin: models that each create their own streams wrap that
    pool after a few instances, two roles then share one HIP stream handle and handle-keyed state (NO_FORK_STREAMS, the per-stream scratch
    buffers) changes the launch order from one model to the next.  A role keeps one stream for the life of the process instead."""
    dev = torch.device(device)
    key = (name, dev.index if dev.index is not None else torch.cuda.current_device(), priority)
    st = _named_streams.get(key)
    if st is None:
        st = _named_streams[key] = torch.cuda.Stream(device=dev, priority=priority)
    return st


PREP_SKIP = os.environ.get('HV_PREP_SKIP', '1') != '0'      # A/B knob: ParamSet.prep(only_if_stale=True) may skip (see there)
LEAN_TABLES = os.environ.get('HV_LEAN_TABLES', '1') != '0' and os.environ.get('HV_WPREP_FUSED', '1') != '0'   # A/B knob: see ParamSet.prep
LEAN = False              # set by lean_tables(): the caller vouches that every convolution of its networks has run once for the current shapes


@contextlib.contextmanager
def lean_tables(on=True):
    """Inside: ParamSet.prep writes only the weight tables the layers' convolutions have been seen to read (ConvParams.use_fwd / use_bwd).  For
    the train step after its first eager step for a batch shape: the same kernels are dispatched again (dispatch is a function of the shapes),
    so the unread tables may go stale.  Outside (eval forwards, the nn.Module API, another shape's first step) every table is written."""
    global LEAN
    prev, LEAN = LEAN, bool(on)
    try:
        yield
    finally:
        LEAN = prev


SPLIT_CONCAT = os.environ.get('HV_SPLIT_CONCAT', '1') != '0'   # [up-sampled | 1 channel] concat layers read the small tensor + the channel (A/B knob)
FUSE_DBIAS = os.environ.get('HV_FUSE_DBIAS', '1') != '0'   # bias gradients computed inside the weight-gradient kernels
SERIAL = False            # True: no side streams at all (per-kernel timing with HIP events needs the GPU to itself)


class GradBook:
    """Gradient twins of activation buffers; first write assigns, later writes accumulate."""

    def __init__(self):
        self.twins = {}
        self.written = set()
        self.side = None          # side HIP stream for weight gradients (they overlap the data-gradient chain)
        self._side_of = None      # ... of this parent stream
        # block deferral (round 4): while `defer_wgrad` is set, conv_backward queues its weight gradient here instead of launching it; the owner of the
        # plan launches the whole block later on ONE side stream (one fork, one join) beside an independent part of the backward
        self.defer_wgrad = False
        self.deferred = []

    def fork(self):
        """Side stream, ordered after everything queued so far on the current stream."""
        cur = torch.cuda.current_stream()
        if self.side is None or self._side_of != cur.cuda_stream:
            self.side, self._side_of = named_stream('wgrad-side-of-%x' % cur.cuda_stream, cur.device), cur.cuda_stream
        self.side.wait_stream(cur)
        return self.side

    def join(self):
        """The current stream waits for the side stream's weight gradients."""
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def twin(self, a):
        t = self.twins.get(id(a.t))
        if t is None:
            t = torch.zeros_like(a.t)
            self.twins[id(a.t)] = t
        return Act(t, a.C, a.coff)

    def reset(self):
        self.written.clear()

    def mark(self, a):
        """-> accumulate flag for a write into the gradient of view `a`."""
        key = (id(a.t), a.coff, a.C)
        acc = key in self.written
        self.written.add(key)
        return acc


def _wgrad(node, p, xin, gfull, accumulate, prec, dbias=None, dbias_accumulate=False):
    # the split-K slab fold of this weight gradient rides in the NEXT weight gradient of the network on this stream (ops.FoldChain; the last one is
    # launched by ParamSet.finish_backward / the owner of a side stream) -- round 4's 62 fold launches per step were each a dependent 5-us node
    chain = p.owner.fold_chain() if p.owner is not None else None
    if node.transposed:
        # y = conv_transpose(x): the weight gradient is that of a strided conv with the roles of x and g swapped;
        # the result is laid out [cin][taps][coutP] (hv_weight_prep_backward knows, transposed_src)
        ops.conv2d_wgrad(gfull, xin, p.dw, node.k, node.s, node.pad, node.d, accumulate=accumulate, precision=prec, chain=chain)
    else:
        ops.conv2d_wgrad(xin, gfull, p.dw, node.k, node.s, node.pad, node.d, in_shift=node.shift, accumulate=accumulate, precision=prec,
                         dbias=dbias, dbias_accumulate=dbias_accumulate, chain=chain)


def conv_backward(node, book, prec, dbias_accumulate=False, wgrad_accumulate=False, tmp_full=None, wgrad=True, x_wg=None, premultiplied=False,
                  mul_x=None, dbias_done=False, bn=None):
    """Backward of one ConvNode: activation gradient (+bias gradient), weight gradient, data gradient.
    x_wg: channel-padded copy of the input for the weight-gradient kernel (1-channel image inputs).
    premultiplied: the gradient buffer of node.y already holds the PRE-activation gradient (its only writer applied act').
    mul_x: activation name of the layer that produced node.x -- the data gradient is multiplied by act'(node.x) in the conv
    epilogue, so that layer's backward starts `premultiplied` (conv_backward_chain).
    bn: (Act raw input of the batch normalisation that produced node.x, its stats, groups, partials) -- the normalisation's backward sums leave
    this data gradient's epilogue (ops.conv2d bn=; the caller asked conv2d_bstats_parts first)."""
    p = node.p
    # the batch-norm sums ride in the plain data gradient only: a caller that set them up for a transposed / pooled / shifted node would skip its reduction
    # pass and read partials nobody wrote
    assert bn is None or (node.need_dx and not node.transposed and node.pool_to is None and not node.shift), 'conv_backward: bn= needs the plain data-gradient branch'
    gy = book.twin(node.y)
    want_dbias = p.bias is not None and node.use_bias and wgrad and not dbias_done      # dbias_done: the caller's seed pass already summed it
    # the bias gradient (column sums of the activation gradient) rides in the weight-gradient kernels, which stream g anyway;
    # conv_transpose nodes (roles of x and g swapped there) and unpadded channel counts keep the stand-alone reduction
    fuse_dbias = want_dbias and not node.transposed and FUSE_DBIAS and p.coutP == p.cout
    if (node.act != 'none' and not premultiplied) or (want_dbias and not fuse_dbias):
        ops.act_backward(gy, node.y, 'none' if premultiplied else node.act, dbias=p.bias.grad if (want_dbias and not fuse_dbias) else None, dbias_accumulate=dbias_accumulate)
    gfull = Act(gy.t, p.coutP, gy.coff)
    if wgrad:
        xs = node.x if x_wg is None else x_wg
        xin = Act(xs.t, p.cin_wg, xs.coff)
        # the weight gradient only feeds the optimiser: queue it on the side stream so that it overlaps the data-gradient
        # chain (joined by GradBook.join() before the gradients are finalised / the activations are overwritten)
        if book.defer_wgrad:      # (its operands -- the layer's input and the finished gradient of its output -- stay as they are until the next backward)
            book.deferred.append(lambda node=node, p=p, xin=xin, gfull=gfull, acc=wgrad_accumulate, db=p.bias.grad if fuse_dbias else None, dba=dbias_accumulate:
                                 _wgrad(node, p, xin, gfull, acc, prec, dbias=db, dbias_accumulate=dba))
        else:
            # (in line: per-layer forks to a side stream lost their A/B twice -- 8.25 -> 8.15 ms at the end of round 3, 7.36 -> 7.6-7.8 ms in round 4 -- and are gone;
            # the refinement generator's weight gradients go to a side stream as ONE block, see defer_wgrad)
            _wgrad(node, p, xin, gfull, wgrad_accumulate, prec, dbias=p.bias.grad if fuse_dbias else None, dbias_accumulate=dbias_accumulate)
    if node.need_dx and node.transposed:
        gx = book.twin(node.x)
        gx = Act(gx.t, p.cin_fwd, gx.coff)
        ops.conv2d(gfull, p.w_bwd, gx, node.k, node.s, node.pad, node.d, transposed=False, accumulate=int(book.mark(gx)), precision=prec, w_h=p.w_bwd_h, w_t=p.w_bwd_t, wuse=(p, 'use_bwd'))
    elif node.need_dx:
        gx = book.twin(node.x)
        gx = Act(gx.t, node.dx_c or p.cin_fwd, gx.coff)
        if node.pool_to is not None:
            # node.x is a concat buffer whose first channels are the nearest x2 up-sampling of pool_to[0]: that part of the data gradient is written
            # 2x2-pooled straight into the small tensor's gradient, times act' of the small tensor (the caller runs its producer premultiplied)
            low, low_act = node.pool_to
            gl = book.twin(low)
            gl = Act(gl.t, node.dx_c or low.C, gl.coff)
            ops.conv2d(gfull, p.w_bwd, gl, node.k, node.s, node.pad, node.d, transposed=True, pool2=True, accumulate=int(book.mark(gl)), precision=prec,
                       w_h=p.w_bwd_h, w_t=p.w_bwd_t, mul=(Act(low.t, gl.C, low.coff), low_act) if low_act != 'none' else None, wuse=(p, 'use_bwd'))
        elif node.shift and _pool2_node_ok(node, gfull, gx, prec):
            # fused up-sampling in the forward: the data gradient leaves 2x2 sum-pooled from the conv's own epilogue (no full-resolution gradient in
            # memory, no hv_copy_channels mode 3 pass), times act'(node.x) when the chain asks for it
            ops.conv2d(gfull, p.w_bwd, gx, node.k, node.s, node.pad, node.d, transposed=True, pool2=True, accumulate=int(book.mark(gx)), precision=prec,
                       w_h=p.w_bwd_h, w_t=p.w_bwd_t, mul=(Act(node.x.t, p.cin_fwd, node.x.coff), mul_x) if mul_x else None, wuse=(p, 'use_bwd'))
        elif node.shift:
            assert not mul_x
            full = tmp_full() if callable(tmp_full) else tmp_full      # (the full-resolution buffer is only built where this fallback runs)
            ops.conv2d(gfull, p.w_bwd, full, node.k, node.s, node.pad, node.d, transposed=True, precision=prec, w_h=p.w_bwd_h, w_t=p.w_bwd_t, wuse=(p, 'use_bwd'))
            ops.copy_channels(full, gx, mode=3, accumulate=book.mark(gx))
        else:
            ops.conv2d(gfull, p.w_bwd, gx, node.k, node.s, node.pad, node.d, transposed=True, accumulate=int(book.mark(gx)), w_h=p.w_bwd_h, w_t=p.w_bwd_t,
                       precision=prec, mul=(Act(node.x.t, p.cin_fwd, node.x.coff), mul_x) if mul_x else None, wuse=(p, 'use_bwd'), bn=bn)


BRANCHES = os.environ.get('HV_G_BRANCHES', '1') != '0'   # independent generator branches on two HIP streams / graph branches
#   (step-level A/B under graph replay, three pairs in one call: 14.80 / 14.81 / 14.85 ms on vs 15.01 / 14.96 / 15.05 ms off)
_branch_streams = {}


def branch_stream():
    """Side stream for an independent branch of the generator (None when everything must stay on one stream).  Weight gradients
    issued on it stay in line (no nested fork: see NO_FORK_STREAMS)."""
    if not BRANCHES or SERIAL:
        return None
    cur = torch.cuda.current_stream()
    if cur.cuda_stream in NO_FORK_STREAMS:
        return None
    key = cur.device.index
    st = _branch_streams.get(key)
    if st is None:
        st = _branch_streams[key] = named_stream('generator-branch', cur.device)
        NO_FORK_STREAMS.add(st.cuda_stream)
    return st


FUSE_ACT = os.environ.get('HV_FUSE_ACT', '1') != '0'   # act' of the producer layer applied in the consumer's data-gradient epilogue


def _pool2_node_ok(node, gfull, gx, prec):
    p = node.p
    return ops.pool2_ok(gfull, gx, node.k, node.s, node.pad, node.d, prec, p.w_bwd_h, p.w_bwd_t, w=p.w_bwd)


def chain_link(n, nxt, prec=None):
    """True when n's data gradient can carry act' of nxt (the producer of n.x): see conv_backward_chain.  A node with fused up-sampling links when
    its data gradient can leave pooled (conv_backward)."""
    if n.shift and nxt is not None and n.need_dx and not n.transposed:
        p = n.p
        gy = Act(n.y.t, p.coutP, n.y.coff)
        if prec is None or not _pool2_node_ok(n, gy, Act(n.x.t, p.cin_fwd, n.x.coff), prec):
            return False
    elif n.shift:
        return False
    return bool(FUSE_ACT and nxt is not None and n.need_dx and not n.transposed and nxt.act != 'none'
                and n.x.t is nxt.y.t and n.x.coff == nxt.y.coff and n.p.cin_fwd <= nxt.y.t.shape[-1] - nxt.y.coff)


def conv_backward_chain(nodes, book, prec, tmp_full=None, premultiplied_first=False, stop_before=None):
    """Backward of a PURE chain of ConvNodes given in backward order: nodes[i].x is exactly the output buffer of nodes[i+1] and
    nothing else reads or writes that buffer's gradient.  Inside the chain the data gradient of nodes[i] is multiplied by
    act'(output of nodes[i+1]) in its conv epilogue, so nodes[i+1] starts from its pre-activation gradient: the in-place
    act-gradient pass (read g, read y, write g) between two convs disappears.
    stop_before: the node that follows nodes[-1] in the chain but is run by the caller later (with
    premultiplied=chain_link(nodes[-1], stop_before))."""
    pre = premultiplied_first and FUSE_ACT     # every writer of nodes[0]'s output gradient already applied its act'
    for i, n in enumerate(nodes):
        nxt = nodes[i + 1] if i + 1 < len(nodes) else stop_before
        link = chain_link(n, nxt, prec)
        conv_backward(n, book, prec, premultiplied=pre, mul_x=nxt.act if link else None, tmp_full=(tmp_full or {}).get(id(n)))
        pre = link


# ================================================================================================ contextual attention
CA_FUSE_PREP = os.environ.get('HV_CA_FUSE_PREP', '1') != '0'   # score-fusion adjoint + Gs + coef in one kernel at 32 x 32 (A/B knob)
CA_F16_IO = os.environ.get('HV_CA_F16_IO', '1') != '0'   # the attention block's boundary kernels read / write fp16-stored maps themselves (A/B knob)
CA_F16_COPIES = os.environ.get('HV_CA_F16_COPIES', '1') != '0'   # GEMM route: fp16 operand copies written by their producers (wp_h, A as fp16 only); A/B knob
CA_GRAM = os.environ.get('HV_CA_GRAM', '1') != '0'     # GEMM route: matching scores and their gradient on the pixel Gram matrix (csrc/attention_gram.hip); A/B knob
CA_GEMM = os.environ.get('HV_CA_GEMM', '1') != '0'     # fp16 mode: the attention block's five contractions as batched NT GEMMs (csrc/bgemm.hip)


def _bgemm(A, B, C, M, N, K, batch, alpha=1.0, colscale=None, b_split=0):
    """C[b] = alpha * colscale[b] (.) A[b] @ B[b]^T over contiguous per-sample matrices A [batch][M][K], B [batch][N][K] (fp32 or fp16),
    C [batch][M][N] fp32 (or fp16: hv_bgemm_nt_h); b_split: B's rows taken in (outer, inner = b_split) order (hv_bgemm_nt)."""
    _lib.get().call('hv_bgemm_nt_h' if C.dtype == torch.float16 else 'hv_bgemm_nt', ptr(A), int(A.dtype == torch.float16), K, ctypes.c_longlong(M * K), ptr(B), int(B.dtype == torch.float16), K,
                    ctypes.c_longlong(N * K), ptr(C), N, ctypes.c_longlong(M * N), M, N, K, batch, ctypes.c_float(alpha), ptr(colscale),
                    ctypes.c_longlong(N if colscale is not None else 0), int(b_split), stream())


class AttentionPlan:
    """ContextualAttention(ksize=3, stride=1, rate=2, fuse_k=3, softmax_scale=10, fuse=True) on an NHWC feature map
    (reference models/inpaint_networks.py:235-410)."""

    def __init__(self, B, H, W, C, device, img_hw, scale=10.0, fuse=True):
        self.B, self.H, self.W, self.C = B, H, W, C
        self.h, self.w = H // 2, W // 2
        assert self.h == self.w and H % 2 == 0, "contextual attention expects a square, even-sized feature map"
        self.L = L = self.h * self.w
        self.img_hw, self.scale, self.fuse = img_hw, scale, fuse
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.fd = Act(z(B, self.h, self.w, C))
        self.wp, self.wpT = z(B, L, 9 * C), z(B, 9 * C, L)
        self.norm, self.rnorm = z(B, L), z(B, L)
        self.raw, self.rawT = z(B, L, 16 * C), z(B, C, 16 * L)
        self.mm = z(L)
        self.mm_b = None      # [B][L], allocated on the first per-sample-mask forward
        self.S0, self.S1, self.A = Act(z(B, self.h, self.w, L)), Act(z(B, self.h, self.w, L)), Act(z(B, self.h, self.w, L))
        self.argmax = torch.zeros(B * L, dtype=torch.int32, device=device)
        self.bw = None

    def forward(self, f, mask_img, out, prec, want_argmax=False, per_sample_mask=False):
        """f: Act [B,H,W,C] (foreground == background), mask_img: (B,1,Himg,Wimg) tensor, out: Act [B,H,W,C].
        per_sample_mask: every sample's own mask decides its valid patches -- the batch stands for B independent batch-1 calls (the
        reference's inference loop); False = the reference's batched behaviour (sample 0's mask for all, inpaint_networks.py:314)."""
        L_ = _lib.get()
        B, H, W, C, L = self.B, self.H, self.W, self.C, self.L
        # the attention block keeps fp32 internally (its score matrices feed a x10 soft-max): fp16-stored feature maps are converted
        # at its boundary (two small copies of the 64-channel map)
        out_user = None
        gemm = CA_GEMM and ops.precision_id(prec) == ops.F16 and (9 * C) % 32 == 0 and L % 32 == 0 and C % 4 == 0
        self.gemm = gemm
        # (the GEMM route's boundary kernels read / write fp16-stored maps themselves: same values, no conversion copies)
        if f.f16 and not (gemm and CA_F16_IO):
            self.f32 = getattr(self, 'f32', None) or Act(torch.zeros(B, H, W, C, dtype=torch.float32, device=f.t.device))
            ops.copy_channels(f, self.f32, mode=0)
            f = self.f32
        if out.f16 and not (gemm and CA_F16_IO):
            self.out32 = getattr(self, 'out32', None) or Act(torch.zeros(B, H, W, C, dtype=torch.float32, device=f.t.device))
            out_user, out = out, self.out32
        if gemm:
            # GEMM route: the patch tables that are only GEMM operands are stored as fp16 (half the bytes through the vector memory path, no
            # conversion when staged); wp stays fp32 (norms, the patch gradient's coefficient term), its transpose is written as fp16
            # Round 3: the copies are written by their PRODUCERS -- wp_h beside wp by the patch kernel, the attention matrix as fp16 only by the
            # soft-max (HV_CA_F16_COPIES=0: the previous form, fp32 tables converted when a GEMM stages them: same GEMM bits) -- and the transpose of
            # wp, which only the backward reads, is taken there
            if getattr(self, 'raw_h', None) is None:
                hz = lambda *s: torch.zeros(*s, dtype=torch.float16, device=f.t.device)
                self.raw_h, self.rawT_h, self.wpT_h = hz(B, L, 16 * C), hz(B, 16 * C, L), hz(B, 9 * C, L)
                self.O = torch.zeros(B, L, 16 * C, dtype=torch.float32, device=f.t.device)
                if CA_F16_COPIES:
                    self.wp_h, self.A_h = hz(B, L, 9 * C), hz(B, L, L)
                    self.O = hz(B, L, 16 * C)          # the paste product only feeds the fold (whose result is stored as fp16): fp16 too
            self.f16_copies = CA_F16_COPIES
            # scores on the pixel Gram matrix (K = C, no patch tables) where the kernels serve the shape
            self.gram = bool(CA_GRAM and self.f16_copies and f.f16 and C == 64 and self.w in (32, 64) and f.ld % 8 == 0 and f.coff == 0)
            if self.gram:
                if getattr(self, 'fd_h', None) is None:
                    self.fd_h = torch.zeros(B, L, C, dtype=torch.float16, device=f.t.device)
                    self.fdT_h = torch.zeros(B, C, L, dtype=torch.float16, device=f.t.device)
                    self.q = torch.zeros(B, L, dtype=torch.float32, device=f.t.device)
                L_.call('hv_ca_gram_down', ptr(f.t), f.f16, B, H, W, C, f.ld, ptr(self.fd_h), ptr(self.fdT_h), ptr(self.q), stream())
            elif self.f16_copies:
                L_.call('hv_ca_patches_h', ptr(f.t), f.f16, B, H, W, C, f.ld, ptr(self.fd.t), ptr(self.wp), ptr(self.wp_h), ptr(self.norm), ptr(self.rnorm), stream())
            else:
                L_.call('hv_ca_patches', ptr(f.t), f.f16, B, H, W, C, f.ld, ptr(self.fd.t), ptr(self.wp), None, ptr(self.norm), ptr(self.rnorm), stream())
                L_.call('hv_transpose_batched_f16', ptr(self.wp), ptr(self.wpT_h), B, L, 9 * C, stream())
            L_.call('hv_ca_raw_patches_f16', ptr(f.t), f.f16, B, H, W, C, f.ld, ptr(self.raw_h), ptr(self.rawT_h), stream())
        else:
            L_.call('hv_ca_patches', ptr(f.t), f.f16, B, H, W, C, f.ld, ptr(self.fd.t), ptr(self.wp), ptr(self.wpT), ptr(self.norm), ptr(self.rnorm), stream())
            L_.call('hv_ca_raw_patches', ptr(f.t), B, H, W, C, f.ld, ptr(self.raw), ptr(self.rawT), stream())
        if per_sample_mask:
            if self.mm_b is None:
                self.mm_b = torch.zeros(B, L, dtype=torch.float32, device=self.mm.device)
            L_.call('hv_ca_mask_batched', ptr(mask_img), B, ctypes.c_longlong(self.img_hw[0] * self.img_hw[1]), self.img_hw[0], self.img_hw[1],
                    self.h, self.w, ptr(self.mm_b), stream())
        else:
            L_.call('hv_ca_mask', ptr(mask_img), self.img_hw[0], self.img_hw[1], self.h, self.w, ptr(self.mm), stream())
        f16c = gemm and self.f16_copies
        if gemm and self.gram:
            L_.call('hv_ca_gram_scores', ptr(self.fd_h), ptr(self.q), B, self.h, self.w, C, ptr(self.S0.t), ptr(self.norm), ptr(self.rnorm), stream())
        elif gemm:    # the 3x3 patches of the (zero-padded) map are both the conv's input columns and its filters: scores = rnorm (.) wp wp^T
            wp_op = self.wp_h if f16c else self.wp
            _bgemm(wp_op, wp_op, self.S0.t, L, L, 9 * C, B, colscale=self.rnorm)
        else:
            ops.conv2d(self.fd, self.wp, self.S0, 3, 1, 1, 1, w_bstride=L * 9 * C, ch_scale=self.rnorm, ch_scale_bstride=L, precision=prec)
        if self.fuse:
            L_.call('hv_ca_fuse', ptr(self.S0.t), ptr(self.S1.t), B, self.h, self.w, 0, stream())
            s = self.S1
        else:
            s = self.S0
        if f16c:
            L_.call('hv_ca_softmax_f16', ptr(s.t), ptr(self.mm_b if per_sample_mask else self.mm), ctypes.c_longlong(L if per_sample_mask else 0),
                    ptr(self.A_h), B, L, ctypes.c_float(self.scale), ptr(self.argmax) if want_argmax else None, stream())
        elif per_sample_mask:
            L_.call('hv_ca_softmax_batched', ptr(s.t), ptr(self.mm_b), ctypes.c_longlong(L), ptr(self.A.t), B, L, ctypes.c_float(self.scale),
                    ptr(self.argmax) if want_argmax else None, stream())
        else:
            L_.call('hv_ca_softmax', ptr(s.t), ptr(self.mm), ptr(self.A.t), B, L, ctypes.c_float(self.scale),
                    ptr(self.argmax) if want_argmax else None, stream())
        if gemm:    # paste = (A rawT^T) folded: O[p][(c, tap)], then every output pixel sums the 4 taps that reach it
            _bgemm(self.A_h if f16c else self.A.t, self.rawT_h, self.O, L, 16 * C, L, B, b_split=C)          # rows of rawT [c][tap] taken as (tap, c): O[p][tap][c]
            L_.call('hv_ca_fold_h' if self.O.dtype == torch.float16 else 'hv_ca_fold', ptr(self.O), ptr(out.t), out.f16, B, H, W, C, out.ld, ctypes.c_float(0.25), 0, stream())
        else:
            ops.conv2d(self.A, self.rawT, out, 4, 2, 1, 1, transposed=True, alpha=0.25, w_bstride=C * 16 * L, precision=prec)
        if out_user is not None:
            ops.copy_channels(out, out_user, mode=0)

    def backward(self, dout, df, accumulate, prec):
        """dout: Act grad of the output; df: Act grad of the input feature map (assigned or accumulated)."""
        L_ = _lib.get()
        B, H, W, C, L = self.B, self.H, self.W, self.C, self.L
        if self.bw is None:
            z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dout.t.device)
            self.bw = dict(dA=Act(z(B, self.h, self.w, L)), AT=Act(z(B, self.h, self.w, L)), dOrawT=z(B, C, 16 * L),
                           dS1=Act(z(B, self.h, self.w, L)), dS0=Act(z(B, self.h, self.w, L)), Gs=Act(z(B, self.h, self.w, L)),
                           coef=z(33 * B, L), dwp=Act(z(B, self.h, self.w, 9 * C)))
        bw = self.bw
        df_user = None
        gemm = getattr(self, 'gemm', False) and ops.precision_id(prec) == ops.F16
        if dout.f16 and not (gemm and CA_F16_IO):
            self.dout32 = getattr(self, 'dout32', None) or Act(torch.zeros(B, H, W, C, dtype=torch.float32, device=dout.t.device))
            ops.copy_channels(dout, self.dout32, mode=0)
            dout = self.dout32
        if df.f16:      # (two kernels add into df: it stays fp32 until both are in, one rounding)
            self.df32 = getattr(self, 'df32', None) or Act(torch.zeros(B, H, W, C, dtype=torch.float32, device=dout.t.device))
            df_user, df, df_acc, accumulate = df, self.df32, accumulate, False
        # through the paste: dA and d(raw patches)
        if gemm:
            if 'dOraw_h' not in bw:
                hz = lambda *s: torch.zeros(*s, dtype=torch.float16, device=dout.t.device)
                bw['dOraw_h'], bw['dOrawT_h'], bw['AT_h'] = hz(B, L, 16 * C), hz(B, 16 * C, L), hz(B, L, L)
            L_.call('hv_ca_raw_patches_f16', ptr(dout.t), dout.f16, B, H, W, C, dout.ld, ptr(bw['dOraw_h']), ptr(bw['dOrawT_h']), stream())
            _bgemm(bw['dOraw_h'], self.raw_h, bw['dA'].t, L, L, 16 * C, B, alpha=0.25)
            if self.f16_copies:
                L_.call('hv_transpose_batched_h2h', ptr(self.A_h), ptr(bw['AT_h']), B, L, L, stream())
            else:
                L_.call('hv_transpose_batched_f16', ptr(self.A.t), ptr(bw['AT_h']), B, L, L, stream())
            _bgemm(bw['AT_h'], bw['dOrawT_h'], self.O, L, 16 * C, L, B, b_split=C)      # d(raw patches)[l][tap][c] (the forward's O buffer is free by now)
            L_.call('hv_ca_fold_h' if self.O.dtype == torch.float16 else 'hv_ca_fold', ptr(self.O), ptr(df.t), df.f16, B, H, W, C, df.ld, ctypes.c_float(0.25),
                    int(bool(accumulate)), stream())
        else:
            ops.conv2d(dout, self.raw, bw['dA'], 4, 2, 1, 1, alpha=0.25, w_bstride=L * 16 * C, precision=prec)
            L_.call('hv_transpose_batched', ptr(self.A.t), ptr(bw['AT'].t), B, L, L, stream())
            L_.call('hv_ca_raw_patches', ptr(dout.t), B, H, W, C, dout.ld, None, ptr(bw['dOrawT']), stream())
            ops.conv2d(bw['AT'], bw['dOrawT'], df, 4, 2, 1, 1, transposed=True, alpha=0.25, w_bstride=C * 16 * L,
                       accumulate=int(accumulate), precision=prec)
        # through softmax and score fusion
        if gemm and self.f16_copies:
            L_.call('hv_ca_softmax_backward_f16', ptr(bw['dA'].t), ptr(self.A_h), ptr(self.mm), ptr(bw['dS1'].t), B, L, ctypes.c_float(self.scale), stream())
        else:
            L_.call('hv_ca_softmax_backward', ptr(bw['dA'].t), ptr(self.A.t), ptr(self.mm), ptr(bw['dS1'].t), B, L, ctypes.c_float(self.scale), stream())
        # ... and through the normalised patch matching (patches act as both filters and inputs)
        use_gram = gemm and getattr(self, 'gram', False)
        if self.fuse and self.h == 32 and self.w == 32 and CA_FUSE_PREP:      # one pass, the plain scores' gradient stays on chip
            L_.call('hv_ca_fuse_backward_prep', ptr(bw['dS1'].t), ptr(self.S0.t), ptr(self.norm), ptr(self.rnorm), ptr(bw['Gs'].t), ptr(bw['coef']),
                    B, self.h, self.w, stream())
        else:
            if self.fuse:
                L_.call('hv_ca_fuse', ptr(bw['dS1'].t), ptr(bw['dS0'].t), B, self.h, self.w, 1, stream())
                ds0 = bw['dS0']
            else:
                ds0 = bw['dS1']
            L_.call('hv_ca_score_backward_prep', ptr(ds0.t), ptr(self.S0.t), ptr(self.norm), ptr(self.rnorm), ptr(bw['Gs'].t), ptr(bw['coef']), B, L, stream())
        if use_gram:
            # d fd = box(Gs) fd + (3x3 sum of coef) fd, added to the even positions of df: one K = L product instead of the L x 9C GEMM + col2im
            L_.call('hv_ca_gram_backward', ptr(bw['Gs'].t), ptr(self.fd_h), ptr(self.fdT_h), ptr(bw['coef']), B, self.h, self.w, C, ptr(df.t), df.ld, stream())
        else:
            if gemm:
                if self.f16_copies:      # (the transpose of wp has this one reader)
                    L_.call('hv_transpose_batched_f16', ptr(self.wp), ptr(self.wpT_h), B, L, 9 * C, stream())
                _bgemm(bw['Gs'].t, self.wpT_h, bw['dwp'].t, L, 9 * C, L, B)
            else:
                ops.conv2d(bw['Gs'], self.wpT, bw['dwp'], 1, 1, 0, 1, w_bstride=9 * C * L, precision=prec)
            L_.call('hv_ca_patches_backward', ptr(bw['dwp'].t), ptr(self.wp), ptr(bw['coef']), ptr(df.t), B, H, W, C, df.ld, 1, stream())
        if df_user is not None:
            ops.copy_channels(df, df_user, mode=0, accumulate=bool(df_acc))
