"""Two-stage inpainting generator on hand-written gfx950 kernels.

API mirror of the reference `models/inpaint_networks.py` (same class names, constructor signatures,
state-dict keys and 7-tuple return: reference :16-32, :36-117, :120-232, :235-410, :413-503), but
the forward/backward are explicit kernel sequences from `engine.py` over NHWC buffers; the torch
modules created here are parameter containers only (their own forward is never run).
"""
import contextlib
import ctypes

import torch
import torch.nn as nn
from torch.nn.utils import spectral_norm as _sn_register

from ._backend import engine as E
from ._backend import lib as _lib
from ._backend import ops
from ._backend import lib as _lib_
ptr, stream = _lib_.ptr, _lib_.stream
from ._backend import ops as _ops_

import os as _os_
HEAD_SEED = _os_.environ.get('HV_HEAD_SEED', '1') != '0'      # fused seed pass of the 1-channel heads (A/B knob)
Act, rup = _ops_.Act, _ops_.rup

_ACTS = ('relu', 'elu', 'lrelu', 'prelu', 'selu', 'tanh', 'sigmoid', 'none')


class Conv2dBlock(nn.Module):
    """Parameter container with the reference's keys: conv.bias, conv.weight_orig, conv.weight_u, conv.weight_v
    (reference :420-503).  Only the configuration the generator uses is executable on the HIP path:
    zero padding, spectral norm, no norm layer, activation in {elu, relu, sigmoid, none}."""

    def __init__(self, input_dim, output_dim, kernel_size, stride, padding=0, conv_padding=0, dilation=1, weight_norm='sn',
                 norm='none', activation='relu', pad_type='zero', transpose=False):
        super().__init__()
        assert pad_type in ('reflect', 'replicate', 'zero', 'none'), "Unsupported padding type: {}".format(pad_type)
        assert norm in ('bn', 'in', 'none'), "Unsupported normalization: {}".format(norm)
        assert weight_norm in ('sn', 'wn', 'none'), "Unsupported normalization: {}".format(weight_norm)
        assert activation in _ACTS, "Unsupported activation: {}".format(activation)
        if pad_type != 'zero' or padding != 0 or norm != 'none' or weight_norm != 'sn' or transpose or \
                activation not in ('elu', 'relu', 'sigmoid', 'none'):
            raise NotImplementedError("Conv2dBlock: only the generator's configuration (zero pad, spectral norm, no norm layer, "
                                      "elu/relu/sigmoid/none) has a HIP kernel path")
        self.use_bias = True
        self.activation_name = activation
        self.cin, self.cout, self.k, self.stride, self.conv_padding, self.dilation = input_dim, output_dim, kernel_size, stride, conv_padding, dilation
        # identical parameter creation (and RNG consumption) to nn.Conv2d + spectral_norm; the hook is removed so the
        # power iteration runs in hv_weight_prep instead.
        conv = _sn_register(nn.Conv2d(input_dim, output_dim, kernel_size, stride, padding=conv_padding, dilation=dilation, bias=True))
        for hid, hook in list(conv._forward_pre_hooks.items()):
            if type(hook).__name__ == 'SpectralNorm':
                del conv._forward_pre_hooks[hid]
        if 'weight' in conv.__dict__:
            del conv.__dict__['weight']
        self.conv = conv

    def params(self, name, cin_fwd=None):
        c = self.conv
        return E.ConvParams(name, c.weight_orig, c.bias, self.cin, self.cout, self.k, cin_fwd=cin_fwd, cin_wg=cin_fwd,
                            u=c.weight_u, v=c.weight_v)

    def forward(self, x):
        raise RuntimeError("Conv2dBlock is executed through Generator (explicit HIP kernel sequence), not stand-alone")


def gen_conv(input_dim, output_dim, kernel_size=3, stride=1, padding=0, rate=1, activation='elu'):
    return Conv2dBlock(input_dim, output_dim, kernel_size, stride, conv_padding=padding, dilation=rate, activation=activation)


class CoarseGenerator(nn.Module):
    def __init__(self, input_dim, cnum, use_cuda):
        super().__init__()
        self.use_cuda = use_cuda
        self.cnum = cnum
        self.conv1 = gen_conv(input_dim + 2, cnum, 5, 1, 2)
        self.conv2_downsample = gen_conv(cnum, cnum * 2, 3, 2, 1)
        self.conv3 = gen_conv(cnum * 2, cnum * 2, 3, 1, 1)
        self.conv4_downsample = gen_conv(cnum * 2, cnum * 4, 3, 2, 1)
        self.conv5 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.conv6 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.conv7_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 2, rate=2)
        self.conv8_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 4, rate=4)
        self.conv9_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 8, rate=8)
        self.conv10_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 16, rate=16)
        self.conv11 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.conv12 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.conv20 = gen_conv(cnum * 4 + 1, cnum * 4, 3, 1, 1)
        self.conv13 = gen_conv(cnum * 4, cnum * 2, 3, 1, 1)
        self.conv14 = gen_conv(cnum * 2, cnum * 2, 3, 1, 1)
        self.conv19 = gen_conv(cnum * 2 + 1, cnum * 2, 3, 1, 1)
        self.conv15 = gen_conv(cnum * 2, cnum, 3, 1, 1)
        self.conv16 = gen_conv(cnum, cnum // 2, 3, 1, 1)
        self.conv17 = gen_conv(cnum // 2, input_dim, 3, 1, 1, activation='none')
        self.conv18 = gen_conv(cnum // 2, input_dim, 3, 1, 1, activation='sigmoid')
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.fc_height = nn.Linear(cnum * 4, 1)

    def forward(self, x, mask, CAM, slice_ratio):
        raise RuntimeError("CoarseGenerator runs as part of Generator's fused kernel sequence")


class FineGenerator(nn.Module):
    def __init__(self, input_dim, cnum, use_cuda=True):
        super().__init__()
        self.use_cuda = use_cuda
        self.cnum = cnum
        self.conv1 = gen_conv(input_dim + 3, cnum, 5, 1, 2)
        self.conv2_downsample = gen_conv(cnum, cnum, 3, 2, 1)
        self.conv3 = gen_conv(cnum, cnum * 2, 3, 1, 1)
        self.conv4_downsample = gen_conv(cnum * 2, cnum * 2, 3, 2, 1)
        self.conv5 = gen_conv(cnum * 2, cnum * 4, 3, 1, 1)
        self.conv6 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.conv7_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 2, rate=2)
        self.conv8_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 4, rate=4)
        self.conv9_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 8, rate=8)
        self.conv10_atrous = gen_conv(cnum * 4, cnum * 4, 3, 1, 16, rate=16)
        self.pmconv1 = gen_conv(input_dim + 3, cnum, 5, 1, 2)
        self.pmconv2_downsample = gen_conv(cnum, cnum, 3, 2, 1)
        self.pmconv3 = gen_conv(cnum, cnum * 2, 3, 1, 1)
        self.pmconv4_downsample = gen_conv(cnum * 2, cnum * 4, 3, 2, 1)
        self.pmconv5 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.pmconv6 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1, activation='relu')
        self.contextul_attention = ContextualAttention(self.use_cuda, ksize=3, stride=1, rate=2, fuse_k=3, softmax_scale=10, fuse=True)
        self.pmconv9 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.pmconv10 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.allconv11 = gen_conv(cnum * 8, cnum * 4, 3, 1, 1)
        self.allconv19 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.allconv12 = gen_conv(cnum * 4, cnum * 4, 3, 1, 1)
        self.allconv13 = gen_conv(cnum * 4, cnum * 2, 3, 1, 1)
        self.allconv14 = gen_conv(cnum * 2, cnum * 2, 3, 1, 1)
        self.allconv15 = gen_conv(cnum * 2, cnum, 3, 1, 1)
        self.allconv16 = gen_conv(cnum, cnum // 2, 3, 1, 1)
        self.allconv17 = gen_conv(cnum // 2 + 1, 1, 3, 1, 1, activation='none')
        self.allconv18 = gen_conv(cnum // 2 + 1, 1, 3, 1, 1, activation='sigmoid')
        self.global_pool = nn.AdaptiveAvgPool2d(1)
        self.fc_height = nn.Linear(cnum * 4, 1)

    def forward(self, xin, x_stage1, mask, coarse_seg, slice_ratio):
        raise RuntimeError("FineGenerator runs as part of Generator's fused kernel sequence")


class ContextualAttention(nn.Module):
    """Contextual attention (Yu et al.) with the reference's signature (reference :235-410).  Stand-alone calls take
    NCHW tensors with f is b; inside Generator the NHWC plan is used directly."""

    def __init__(self, use_cuda, ksize=3, stride=1, rate=1, fuse_k=3, softmax_scale=10, fuse=False):
        super().__init__()
        self.ksize, self.stride, self.rate, self.fuse_k, self.softmax_scale, self.fuse, self.use_cuda = \
            ksize, stride, rate, fuse_k, softmax_scale, fuse, use_cuda
        self._plans = {}

    def check(self):
        if not (self.ksize == 3 and self.stride == 1 and self.rate == 2 and self.fuse_k == 3):
            raise NotImplementedError("ContextualAttention HIP path: ksize=3, stride=1, rate=2, fuse_k=3 only")

    def plan(self, B, H, W, C, device, img_hw):
        self.check()
        key = (B, H, W, C, str(device), img_hw)
        if key not in self._plans:
            self._plans[key] = E.AttentionPlan(B, H, W, C, device, img_hw, float(self.softmax_scale), bool(self.fuse))
        return self._plans[key]

    def forward(self, f, b, mask=None):
        if f is not b and not torch.equal(f, b):
            raise NotImplementedError("ContextualAttention HIP path expects foreground and background to be the same tensor")
        _lib.require_gpu(f)
        B, C, H, W = f.shape
        if mask is None:
            mask = torch.zeros(B, 1, 4 * H, 4 * W, device=f.device)
        fa = ops.from_nchw(f.detach())
        out = Act.empty(B, H, W, C, f.device)
        pl = self.plan(B, H, W, C, f.device, (mask.shape[2], mask.shape[3]))
        pl.forward(fa, mask.contiguous().float(), out, ops.default_precision(), want_argmax=True)
        y = out.nchw()
        flow = offsets_to_flow(pl.argmax, B, pl.h, pl.w, self.rate)
        if f.requires_grad and torch.is_grad_enabled():
            return _AttentionFn.apply(f, y, pl, fa), flow
        return y, flow


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f, y, plan, fa):
        ctx.plan, ctx.shape = plan, f.shape
        return y.clone()

    @staticmethod
    def backward(ctx, gy):
        B, C, H, W = ctx.shape
        g = ops.from_nchw(gy.contiguous())
        df = Act.empty(B, H, W, C, gy.device)
        ctx.plan.backward(g, df, False, ops.default_precision())
        return df.nchw(), None, None, None


def offsets_to_flow(argmax, B, h, w, rate):
    """offset_flow slot of the 7-tuple (reference :368,:389-410): the arg-max offsets coloured with the Middlebury wheel and nearest-
    upsampled by rate*4.  The reference does this with NumPy on the host (inpaint_tools.py:73-100), one device sync per forward; here
    it is one small kernel on the arg-max indices the soft-max already produced (hv_ca_flow)."""
    flow = torch.empty(B, 3, h * rate * 4, w * rate * 4, dtype=torch.float32, device=argmax.device)
    _lib.get().call('hv_ca_flow', ptr(argmax), B, h, w, rate * 4, ptr(flow), stream())
    return flow


# ================================================================================================ generator plan
class _GenPlan:
    """Buffers + node list of Generator for one (B, H, W)."""

    def __init__(self, gen, B, H, W, device):
        cg, fg = gen.coarse_generator, gen.fine_generator
        c = gen.cnum
        self.B, self.H, self.W, self.dev = B, H, W, device
        dt = ops.storage_dtype(gen.precision)        # fp16 buffers in the fp16 mode; the (B,1,H,W) image outputs stay fp32
        z = lambda h, w, C: Act(torch.zeros(B, h, w, ops.cpad(C), dtype=dt, device=device), C, 0)
        img = lambda: torch.zeros(B, 1, H, W, dtype=torch.float32, device=device)
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        self.book = E.GradBook()
        P = gen._pset_convs
        N = E.ConvNode
        # ---------------- coarse
        self.c_in = z(H, W, 3)
        a = {}
        a['c1'] = z(H, W, c); a['c2'] = z(H2, W2, 2 * c); a['c3'] = z(H2, W2, 2 * c); a['c4'] = z(H4, W4, 4 * c)
        for n in ('c5', 'c6', 'c7', 'c8', 'c9', 'c10', 'c11', 'c12'):
            a[n] = z(H4, W4, 4 * c)
        a['cat20'] = z(H2, W2, 4 * c + 1); a['c20'] = z(H2, W2, 4 * c); a['c13'] = z(H2, W2, 2 * c); a['c14'] = z(H2, W2, 2 * c)
        a['cat19'] = z(H, W, 2 * c + 1); a['c19'] = z(H, W, 2 * c); a['c15'] = z(H, W, c); a['c16'] = z(H, W, c // 2)
        self.x_stage1, self.coarse_seg = img(), img()
        xs1 = Act(self.x_stage1.view(B, H, W, 1)); cs = Act(self.coarse_seg.view(B, H, W, 1))
        self.c_nodes = [
            N(P['coarse_generator.conv1'], self.c_in, a['c1'], 1, 2, 1, 'elu', need_dx=False),
            N(P['coarse_generator.conv2_downsample'], a['c1'], a['c2'], 2, 1, 1, 'elu'),
            N(P['coarse_generator.conv3'], a['c2'], a['c3'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv4_downsample'], a['c3'], a['c4'], 2, 1, 1, 'elu'),
            N(P['coarse_generator.conv5'], a['c4'], a['c5'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv6'], a['c5'], a['c6'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv7_atrous'], a['c6'], a['c7'], 1, 2, 2, 'elu'),
            N(P['coarse_generator.conv8_atrous'], a['c7'], a['c8'], 1, 4, 4, 'elu'),
            N(P['coarse_generator.conv9_atrous'], a['c8'], a['c9'], 1, 8, 8, 'elu'),
            N(P['coarse_generator.conv10_atrous'], a['c9'], a['c10'], 1, 16, 16, 'elu'),
            N(P['coarse_generator.conv11'], a['c10'], a['c11'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv12'], a['c11'], a['c12'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv20'], a['cat20'], a['c20'], 1, 1, 1, 'elu', dx_c=4 * c),      # (the CAM channel is an input: no gradient)
            N(P['coarse_generator.conv13'], a['c20'], a['c13'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv14'], a['c13'], a['c14'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv19'], a['cat19'], a['c19'], 1, 1, 1, 'elu', dx_c=2 * c),
            N(P['coarse_generator.conv15'], a['c19'], a['c15'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv16'], a['c15'], a['c16'], 1, 1, 1, 'elu'),
            N(P['coarse_generator.conv17'], a['c16'], xs1, 1, 1, 1, 'clamp'),
            N(P['coarse_generator.conv18'], a['c16'], cs, 1, 1, 1, 'sigmoid'),
        ]
        self.c_nodes[12].split = (a['c12'], a['cat20'].slice(4 * c, 1))      # conv20 = [up(c12) | CAM at 128^2]
        self.c_nodes[15].split = (a['c14'], a['cat19'].slice(2 * c, 1))      # conv19 = [up(c14) | CAM]
        self.c_pool = torch.zeros(B, 4 * c, device=device); self.pred1 = torch.zeros(B, 1, device=device)
        # ---------------- fine
        self.f_in = z(H, W, 4)
        a['f1'] = z(H, W, c); a['f2'] = z(H2, W2, c); a['f3'] = z(H2, W2, 2 * c); a['f4'] = z(H4, W4, 2 * c)
        for n in ('f5', 'f6', 'f7', 'f8', 'f9'):
            a[n] = z(H4, W4, 4 * c)
        a['cat11'] = z(H4, W4, 8 * c)
        a['p1'] = z(H, W, c); a['p2'] = z(H2, W2, c); a['p3'] = z(H2, W2, 2 * c)
        for n in ('p4', 'p5', 'p6', 'ca', 'p9'):
            a[n] = z(H4, W4, 4 * c)
        for n in ('a11', 'a12', 'a19'):
            a[n] = z(H4, W4, 4 * c)
        a['a13'] = z(H2, W2, 2 * c); a['a14'] = z(H2, W2, 2 * c); a['a15'] = z(H, W, c)
        a['cat17'] = z(H, W, c // 2 + 1)
        self.x_stage2, self.fine_seg = img(), img()
        xs2 = Act(self.x_stage2.view(B, H, W, 1)); fs = Act(self.fine_seg.view(B, H, W, 1))
        hallu, pm = a['cat11'].slice(0, 4 * c), a['cat11'].slice(4 * c, 4 * c)
        a16 = a['cat17'].slice(0, c // 2)
        fp = 'fine_generator.'
        self.f_nodes_conv = [
            N(P[fp + 'conv1'], self.f_in, a['f1'], 1, 2, 1, 'elu'),
            N(P[fp + 'conv2_downsample'], a['f1'], a['f2'], 2, 1, 1, 'elu'),
            N(P[fp + 'conv3'], a['f2'], a['f3'], 1, 1, 1, 'elu'),
            N(P[fp + 'conv4_downsample'], a['f3'], a['f4'], 2, 1, 1, 'elu'),
            N(P[fp + 'conv5'], a['f4'], a['f5'], 1, 1, 1, 'elu'),
            N(P[fp + 'conv6'], a['f5'], a['f6'], 1, 1, 1, 'elu'),
            N(P[fp + 'conv7_atrous'], a['f6'], a['f7'], 1, 2, 2, 'elu'),
            N(P[fp + 'conv8_atrous'], a['f7'], a['f8'], 1, 4, 4, 'elu'),
            N(P[fp + 'conv9_atrous'], a['f8'], a['f9'], 1, 8, 8, 'elu'),
            N(P[fp + 'conv10_atrous'], a['f9'], hallu, 1, 16, 16, 'elu'),
        ]
        self.f_nodes_pm = [
            N(P[fp + 'pmconv1'], self.f_in, a['p1'], 1, 2, 1, 'elu'),
            N(P[fp + 'pmconv2_downsample'], a['p1'], a['p2'], 2, 1, 1, 'elu'),
            N(P[fp + 'pmconv3'], a['p2'], a['p3'], 1, 1, 1, 'elu'),
            N(P[fp + 'pmconv4_downsample'], a['p3'], a['p4'], 2, 1, 1, 'elu'),
            N(P[fp + 'pmconv5'], a['p4'], a['p5'], 1, 1, 1, 'elu'),
            N(P[fp + 'pmconv6'], a['p5'], a['p6'], 1, 1, 1, 'relu'),
        ]
        self.f_nodes_pm2 = [
            N(P[fp + 'pmconv9'], a['ca'], a['p9'], 1, 1, 1, 'elu'),
            N(P[fp + 'pmconv10'], a['p9'], pm, 1, 1, 1, 'elu'),
        ]
        self.f_nodes_merge = [
            N(P[fp + 'allconv11'], a['cat11'], a['a11'], 1, 1, 1, 'elu'),
            N(P[fp + 'allconv12'], a['a11'], a['a12'], 1, 1, 1, 'elu'),
            N(P[fp + 'allconv19'], a['a12'], a['a19'], 1, 1, 1, 'elu'),
            N(P[fp + 'allconv13'], a['a19'], a['a13'], 1, 1, 1, 'elu', shift=1),
            N(P[fp + 'allconv14'], a['a13'], a['a14'], 1, 1, 1, 'elu'),
            N(P[fp + 'allconv15'], a['a14'], a['a15'], 1, 1, 1, 'elu', shift=1),
            N(P[fp + 'allconv16'], a['a15'], a16, 1, 1, 1, 'elu'),
            N(P[fp + 'allconv17'], a['cat17'], xs2, 1, 1, 1, 'clamp'),
            N(P[fp + 'allconv18'], a['cat17'], fs, 1, 1, 1, 'sigmoid'),
        ]
        self.f_pool = torch.zeros(B, 4 * c, device=device); self.pred2 = torch.zeros(B, 1, device=device)
        self.a = a
        self.attn = fg.contextul_attention.plan(B, H4, W4, 4 * c, device, (H, W))
        # full-resolution scratch for the data gradients of the two fused-upsample convolutions
        self.tmp_up = {}
        # 4-channel padded carriers for the 1-channel head gradients
        self.g_head = {n: z(H, W, 1) for n in ('c17', 'c18', 'f17', 'f18')}


G_WGRAD_BLOCK = _os_.environ.get('HV_G_WGRAD_BLOCK', '1') != '0'     # A/B knob: see Generator.run_backward
G_WGRAD_COARSE = int(_os_.environ.get('HV_G_WGRAD_COARSE', '2'))      # A/B knob: see Generator.run_backward


class Generator(nn.Module):
    def __init__(self, config, use_cuda):
        super().__init__()
        self.input_dim = config['input_dim']
        self.cnum = config['ngf']
        self.use_cuda = use_cuda
        if self.input_dim != 1:
            raise NotImplementedError("Generator HIP path: input_dim == 1 (single-channel CT slices)")
        self.coarse_generator = CoarseGenerator(self.input_dim, self.cnum, self.use_cuda)
        self.fine_generator = FineGenerator(self.input_dim, self.cnum, self.use_cuda)
        self.precision = None          # None -> HV_PRECISION env (fp32 parity mode by default)
        self._plans = {}
        self._pset = None
        self._eval_graphs = {}
        import os as _os
        self.use_graph = _os.environ.get('HV_GRAPH', '1') != '0'
        self._pset_convs = None
        self._tail_stream = None       # a stream still updating the weights (the data-parallel step's exchange stream)

    def _wait_tail(self):
        if self._tail_stream is not None and not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream().wait_stream(self._tail_stream)

    # ---------------------------------------------------------------- parameters
    def paramset(self):
        if self._pset is None:
            convs = {}
            for gname in ('coarse_generator', 'fine_generator'):
                g = getattr(self, gname)
                for n, m in g.named_children():
                    if isinstance(m, Conv2dBlock):
                        convs['%s.%s' % (gname, n)] = m.params('%s.%s' % (gname, n), cin_fwd=ops.cpad(m.cin))
            # conv19 / conv20 read [up-sampled feature map | CAM]: their first 2c / 4c input channels also as a filter table of their own, so that the
            # forward can read the small map with the fused up-sampling and take the CAM channel in its epilogue (ConvNode.split, hv_conv_desc.x1)
            for n, k2 in (('coarse_generator.conv19', 2 * self.cnum), ('coarse_generator.conv20', 4 * self.cnum)):
                if n in convs and k2 % 32 == 0 and convs[n].cin == k2 + 1:
                    convs[n].split_k = k2
            self._pset_convs = convs
            cg, fg = self.coarse_generator, self.fine_generator
            self._pset = E.ParamSet(convs.values(), [cg.fc_height.weight, cg.fc_height.bias, fg.fc_height.weight, fg.fc_height.bias])
        return self._pset

    def _plan(self, B, H, W, device):
        key = (B, H, W, str(device), ops.precision_id(self.precision))
        if key not in self._plans:
            if H % 8 or W % 8 or H != W:
                raise NotImplementedError("Generator HIP path expects square inputs with a side divisible by 8")
            self.paramset()
            self._plans[key] = _GenPlan(self, B, H, W, device)
        return self._plans[key]

    # ---------------------------------------------------------------- explicit forward / backward
    def run_forward(self, x, mask, CAM, slice_ratio, training=None, per_sample_mask=False):
        """x, mask, CAM: (B,1,H,W) fp32 device tensors; slice_ratio: (B,) fp64.  Returns the plan (all activations
        stay in its buffers); outputs are plan.coarse_seg/fine_seg/x_stage1/x_stage2 (B,1,H,W) and plan.pred1/pred2 (B,1).
        per_sample_mask: the batch stands for B independent batch-1 calls (attention masks per sample, see AttentionPlan.forward)."""
        _lib.require_gpu(x, mask, CAM)
        training = self.training if training is None else training
        prec = ops.precision_id(self.precision)
        B, _, H, W = x.shape
        dev = x.device
        P = self._plan(B, H, W, dev)
        self.paramset().prep(dev, power_iter=training)
        x = x.contiguous().float(); mask = mask.contiguous().float(); CAM = CAM.contiguous().float()
        ratio = slice_ratio.to(device=dev, dtype=torch.float64).contiguous()
        P.mask_img = mask
        cg, fg = self.coarse_generator, self.fine_generator
        c = self.cnum
        a = P.a
        cam = Act(CAM.view(B, H, W, 1))
        # ---- coarse
        ops.gen_input(x, None, mask, ratio, P.c_in, 0)
        for n in P.c_nodes[:10]:
            n.forward(prec)
        ops.gap_fc_sigmoid(a['c10'], cg.fc_height.weight, cg.fc_height.bias, P.c_pool, P.pred1)
        P.c_nodes[10].forward(prec); P.c_nodes[11].forward(prec)
        # (split layers read c12 / c14 themselves: the up-sampled part of the concat buffer is then built by the backward, on its side stream)
        if not P.c_nodes[12].split_forward(prec):
            ops.copy_channels(a['c12'], a['cat20'].slice(0, 4 * c), mode=1)
        ops.copy_channels(cam, a['cat20'].slice(4 * c, 1), mode=2)
        for n in P.c_nodes[12:15]:
            n.forward(prec)
        if not P.c_nodes[15].split_forward(prec):
            ops.copy_channels(a['c14'], a['cat19'].slice(0, 2 * c), mode=1)
        ops.copy_channels(cam, a['cat19'].slice(2 * c, 1), mode=0)
        for n in P.c_nodes[15:]:
            n.forward(prec)
        # ---- fine
        tf = getattr(self, 'time_fine', None)
        if tf is not None and not torch.cuda.is_current_stream_capturing():     # bench.py: HIP events around the refinement generator
            tf.append((torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
            tf[-1][0].record()
        else:
            tf = None
        self._fine_forward(P, x, mask, ratio, prec, per_sample_mask)
        if tf is not None:
            tf[-1][1].record()
        return P

    def _fine_forward(self, P, x, mask, ratio, prec, per_sample_mask=False):
        """FineGenerator.forward (reference models/inpaint_networks.py:169-232) over the plan's buffers: reads x, mask, P.coarse_seg, P.x_stage1."""
        B, _, H, W = x.shape
        fg, c, a = self.fine_generator, self.cnum, P.a
        ops.gen_input(x, P.coarse_seg, mask, ratio, P.f_in, 1)
        # the dilated-conv branch and the attention branch only share their input: two streams (two branches of the step graph)
        side = E.branch_stream()
        main = torch.cuda.current_stream()
        if side is not None:
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            for n in P.f_nodes_pm:
                n.forward(prec)
            P.attn.forward(a['p6'], mask, a['ca'], prec, want_argmax=True, per_sample_mask=per_sample_mask)
            for n in P.f_nodes_pm2:
                n.forward(prec)
        for n in P.f_nodes_conv:
            n.forward(prec)
        if side is not None:
            main.wait_stream(side)
        P.f_nodes_merge[0].forward(prec)
        ops.gap_fc_sigmoid(a['a11'], fg.fc_height.weight, fg.fc_height.bias, P.f_pool, P.pred2)
        for n in P.f_nodes_merge[1:7]:
            n.forward(prec)
        ops.copy_channels(Act(P.x_stage1.view(B, H, W, 1)), a['cat17'].slice(c // 2, 1), mode=0)
        P.f_nodes_merge[7].forward(prec); P.f_nodes_merge[8].forward(prec)

    def fine_forward_graph(self, P, x, mask, slice_ratio):
        """bench.py: the refinement generator's training forward alone as ONE captured hipGraph over the buffers of plan P (which a full
        run_forward has filled: coarse outputs, prepared weight tables) -- both branches on their two streams, as inside the step graphs."""
        prec = ops.precision_id(self.precision)
        ratio = slice_ratio.to(device=x.device, dtype=torch.float64).contiguous()
        x = x.contiguous().float(); mask = mask.contiguous().float()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=E.named_stream('capture', x.device), capture_error_mode='thread_local'):
            self._fine_forward(P, x, mask, ratio, prec)
        self._fine_graph_keep = (x, mask, ratio)
        return g

    def _tmp_up(self, P, node):
        key = id(node)
        if key not in P.tmp_up:
            x = node.x
            P.tmp_up[key] = Act(torch.zeros(x.B, x.H * 2, x.W * 2, x.ld, dtype=x.t.dtype, device=x.t.device), node.p.cin_fwd, 0)
        return P.tmp_up[key]

    def _head_backward(self, P, node, seed, gname, prec, book, mul_x=None):
        """1-channel head: seed (B,1,H,W) -> padded carrier -> activation/bias gradient -> wgrad + dgrad."""
        carrier = P.g_head[gname]                       # [B,H,W,4], channel 0 live
        book.twins[id(node.y.t)] = carrier.t            # the head's output gradient lives in the carrier
        pn = node.p
        if HEAD_SEED and carrier.f16 and seed.dtype == torch.float32 and seed.is_contiguous() and pn.bias is not None and node.use_bias:
            # seed -> act' -> carrier -> bias gradient in one pass (was: copy, in-place act' pass, column sums)
            ops.head_seed_backward(seed, node.y, Act(carrier.t, 4, 0), node.act, dbias=pn.bias.grad)
            E.conv_backward(node, book, prec, premultiplied=True, dbias_done=True, mul_x=mul_x if E.FUSE_ACT else None)
            return
        ops.copy_channels(Act(seed.view(P.B, P.H, P.W, 1)), carrier, mode=0)
        E.conv_backward(node, book, prec, mul_x=mul_x if E.FUSE_ACT else None)

    def run_backward(self, P, d_coarse_seg, d_fine_seg, d_x_stage1, d_x_stage2, d_pred1, d_pred2):
        """Gradients of a scalar loss wrt the six differentiable outputs -> .grad of every parameter.
        All seeds are dense fp32 device tensors shaped like the outputs (None = zero)."""
        prec = ops.precision_id(self.precision)
        B, H, W, c = P.B, P.H, P.W, self.cnum
        book = P.book
        book.reset()
        a = P.a
        cg, fg = self.coarse_generator, self.fine_generator
        zero = lambda t, ref: torch.zeros_like(ref) if t is None else t.contiguous().float()
        d_fine_seg, d_x_stage2 = zero(d_fine_seg, P.fine_seg), zero(d_x_stage2, P.x_stage2)
        d_coarse_seg, d_x_stage1 = zero(d_coarse_seg, P.coarse_seg), zero(d_x_stage1, P.x_stage1)
        d_pred1, d_pred2 = zero(d_pred1, P.pred1), zero(d_pred2, P.pred2)
        M = P.f_nodes_merge
        # the concat inputs of the split layers (forward: never built) for their weight gradients: up-sampled now, beside the head kernels -- or, where the
        # layer's weight gradient goes to the side stream (HV_G_WGRAD_COARSE), on that stream right in front of it: 35 + 20 us of copies (67 + 33 MB written)
        # off the head of the backward's critical path
        late_copies = []
        wg_block_now = G_WGRAD_BLOCK and not E.SERIAL and torch.cuda.current_stream().cuda_stream not in E.NO_FORK_STREAMS
        for grp, (node, low, cat, k2) in enumerate(((P.c_nodes[15], a['c14'], a['cat19'], 2 * c), (P.c_nodes[12], a['c12'], a['cat20'], 4 * c)), start=1):
            if node.split_forward(prec):
                if wg_block_now and G_WGRAD_COARSE >= grp:
                    late_copies.append(lambda low=low, cat=cat, k2=k2: ops.copy_channels(low, cat.slice(0, k2), mode=1))
                else:
                    ops.copy_channels(low, cat.slice(0, k2), mode=1)
        # the refinement generator's weight gradients as ONE block on a side stream beside the coarse generator's whole backward (round 4, HV_G_WGRAD_BLOCK):
        # they only feed the optimiser, and the coarse backward -- a chain of small launches that leave most of a CU's registers and LDS free -- does not
        # depend on them.  One fork and one join (per-layer forks cost more than they returned and are gone).  Measured against it, three
        # same-box pairs each: a first block launched before the two branches (three streams busy there) +0.13 ms; the coarse generator's own weight
        # gradients in two more blocks +0.14 ms -- both removed.
        wg_block = G_WGRAD_BLOCK and not E.SERIAL and torch.cuda.current_stream().cuda_stream not in E.NO_FORK_STREAMS
        book.defer_wgrad = bool(wg_block)
        wg_side = E.named_stream('generator-wgrad-block', d_x_stage2.device) if wg_block else None
        def launch_block():
            # (several side streams with the launches dealt round-robin, HV_G_WGRAD_STREAMS 2 / 3, measured slower in round 5: 6.94 -> 6.98-7.36 ms)
            wg_side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(wg_side):
                for launch in book.deferred:
                    launch()
                self.paramset().fold_chain().flush()      # the side stream's last slab fold, on that stream (before the join)
            book.deferred = []
        # ---- fine: heads
        self._head_backward(P, M[7], d_x_stage2, 'f17', prec, book)
        self._head_backward(P, M[8], d_fine_seg, 'f18', prec, book)
        gcat17 = book.twin(a['cat17'])
        # x_stage1 also feeds the fine heads (channel c/2 of cat17)
        if 'd_xs1_total' not in P.__dict__:      # (setdefault(..., torch.zeros_like(...)) built and filled the default on every call: a fill kernel per step)
            P.d_xs1_total, P.d_cs_total = torch.zeros_like(P.x_stage1), torch.zeros_like(P.coarse_seg)
        d_xs1_total = P.d_xs1_total
        ops.add_channels(Act(d_x_stage1.view(B, H, W, 1)), gcat17.slice(c // 2, 1), Act(d_xs1_total.view(B, H, W, 1)))
        # pure links (single producer, single consumer): the consumer's data gradient applies the producer's act'
        # (M[5] and M[3] read their input up-sampled: they link too where their data gradient can leave pooled, otherwise the chain breaks there)
        E.conv_backward_chain([M[6], M[5], M[4], M[3], M[2], M[1]], book, prec, stop_before=M[0],
                              tmp_full={id(M[5]): lambda: self._tmp_up(P, M[5]), id(M[3]): lambda: self._tmp_up(P, M[3])})
        # a11 (allconv11's output, input of M[1]) also feeds the height head: both writers of its gradient apply elu'(a11)
        pre11 = E.chain_link(M[1], M[0], prec)
        ops.gap_fc_sigmoid_backward(d_pred2, P.pred2, P.f_pool, fg.fc_height.weight, book.twin(a['a11']),
                                    fg.fc_height.weight.grad, fg.fc_height.bias.grad, mul=(a['a11'], M[0].act) if pre11 else None)
        # cat11 = [conv10_atrous | pmconv10], both ELU: allconv11's data gradient applies elu' for both producers
        E.conv_backward(M[0], book, prec, premultiplied=pre11, mul_x='elu' if E.FUSE_ACT else None)
        # the two branches run concurrently; both end in the gradient of f_in: the attention branch stops before its first conv,
        # which is run after the join (assign / accumulate order of the shared buffer stays that of the single-stream schedule)
        side = E.branch_stream()
        main = torch.cuda.current_stream()
        pm_rev = list(reversed(P.f_nodes_pm))
        if side is not None:
            side.wait_stream(main)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            E.conv_backward_chain(list(reversed(P.f_nodes_pm2)), book, prec, premultiplied_first=True)
            gp6 = book.twin(a['p6'])
            P.attn.backward(book.twin(a['ca']), gp6, book.mark(gp6), prec)
            E.conv_backward_chain(pm_rev if side is None else pm_rev[:-1], book, prec, stop_before=None if side is None else pm_rev[-1])
            if side is not None:
                self.paramset().fold_chain().flush()      # (weight gradients issued in line on the branch stream: their last slab fold, before the join)
        E.conv_backward_chain(list(reversed(P.f_nodes_conv)), book, prec, premultiplied_first=True)
        if side is not None:
            main.wait_stream(side)
            E.conv_backward(pm_rev[-1], book, prec, premultiplied=E.chain_link(pm_rev[-2], pm_rev[-1], prec))
        # coarse_seg enters the fine generator as channel 1 of its input
        d_cs_total = P.d_cs_total
        ops.add_channels(Act(d_coarse_seg.view(B, H, W, 1)), book.twin(P.f_in).slice(1, 1), Act(d_cs_total.view(B, H, W, 1)))
        if wg_block:
            book.defer_wgrad = G_WGRAD_COARSE > 0      # (the first coarse layers' weight gradients join the side streams' queue behind this block: see below)
            launch_block()
            book.deferred.extend(late_copies)          # (in front of the coarse weight gradients that read them, on their stream)
        # ---- coarse
        C = P.c_nodes
        # both heads read c16 (output of conv16, ELU): each applies elu'(c16) to its share of the gradient
        self._head_backward(P, C[18], d_xs1_total, 'c17', prec, book, mul_x='elu')
        self._head_backward(P, C[19], d_cs_total, 'c18', prec, book, mul_x='elu')
        # conv19 / conv20 read [up-sampled c14 / c12 | CAM]: where the pooled data gradient serves the shape (fp16 mode) the gradient of the small tensor
        # comes 2x2-pooled and times elu' straight from the conv's epilogue; otherwise full-resolution gradient + adjoint-of-up-sampling pass
        def pooled(node, low):
            pn = node.p
            g = E.Act(book.twin(node.y).t, pn.coutP, node.y.coff)
            ok = E.FUSE_ACT and ops.pool2_ok(g, E.Act(book.twin(low).t, node.dx_c, low.coff), node.k, node.s, node.pad, node.d, prec, pn.w_bwd_h, pn.w_bwd_t)
            node.pool_to = (low, 'elu') if ok else None
            return ok
        p19 = pooled(C[15], a['c14'])
        E.conv_backward_chain([C[17], C[16], C[15]], book, prec, premultiplied_first=True)
        if not p19:
            g14 = book.twin(a['c14'])
            ops.copy_channels(book.twin(a['cat19']).slice(0, 2 * c), g14, mode=3, accumulate=book.mark(g14))
        # HV_G_WGRAD_COARSE (round 5): the coarse generator's first backward layers -- its heads and the 256 x 256 / 128 x 128 decoder, the most expensive weight
        # gradients of the chain -- hand their weight gradients to the side stream too (ONE more fork: it queues them behind the refinement generator's
        # block), so that the main stream walks these layers with data gradients only; the rest of the coarse backward keeps its weight gradients in line
        # (everything on the side stream made the block outlast the chain: +0.14 ms, round 4)
        if wg_block and G_WGRAD_COARSE == 1:
            book.defer_wgrad = False
            launch_block()
        p20 = pooled(C[12], a['c12'])
        E.conv_backward_chain([C[14], C[13], C[12]], book, prec, premultiplied_first=p19)
        if not p20:
            g12 = book.twin(a['c12'])
            ops.copy_channels(book.twin(a['cat20']).slice(0, 4 * c), g12, mode=3, accumulate=book.mark(g12))
        if wg_block and G_WGRAD_COARSE == 2:
            book.defer_wgrad = False
            launch_block()
        E.conv_backward_chain([C[11], C[10]], book, prec, premultiplied_first=p20, stop_before=C[9])
        if wg_block and G_WGRAD_COARSE >= 3:
            book.defer_wgrad = False
            launch_block()
        pre10 = E.chain_link(C[10], C[9], prec)      # c10 feeds conv11 and the height head: both apply elu'(c10)
        ops.gap_fc_sigmoid_backward(d_pred1, P.pred1, P.c_pool, cg.fc_height.weight, book.twin(a['c10']),
                                    cg.fc_height.weight.grad, cg.fc_height.bias.grad, mul=(a['c10'], C[9].act) if pre10 else None)
        E.conv_backward_chain(list(reversed(C[:10])), book, prec, premultiplied_first=pre10)
        book.join()     # side-stream weight gradients
        if wg_side is not None:
            torch.cuda.current_stream().wait_stream(wg_side)
        self.paramset().finish_backward(accumulate=False)
        self.paramset().attach_grads()

    # ---------------------------------------------------------------- nn.Module API
    def forward(self, x, mask, CAM, slice_ratio):
        """Same signature and 7-tuple as the reference (inpaint_networks.py:28-32):
        (coarse_seg, fine_seg, x_stage1, x_stage2, offset_flow, pred1_h, pred2_h)."""
        if not torch.is_tensor(slice_ratio):
            slice_ratio = torch.as_tensor(slice_ratio, dtype=torch.float64).reshape(-1)
        self._wait_tail()
        if self.use_graph and not self.training and not torch.is_grad_enabled() and ops.timer() is None and x.is_cuda:
            P = self._eval_replay(x, mask, CAM, slice_ratio)
        else:
            P = self.run_forward(x, mask, CAM, slice_ratio)
        outs = (P.coarse_seg, P.fine_seg, P.x_stage1, P.x_stage2, P.pred1, P.pred2)
        flow = offsets_to_flow(P.attn.argmax, P.B, P.attn.h, P.attn.w, 2)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            anchor = next(p for p in self.parameters() if p.requires_grad)
            o = _GeneratorFn.apply(anchor, self, P, *outs)
        else:
            o = tuple(t.clone() for t in outs)
        return o[0], o[1], o[2], o[3], flow, o[4], o[5]


def _generator_eval_replay(self, x, mask, CAM, slice_ratio):
    """Eval-mode forward as a captured hipGraph per input shape: the ~250 launches of a bs=1 inference call (launch-bound when
    issued eagerly: the reference's eval loop runs ~130 of them per volume) become one graph launch.  Inputs are copied into
    the graph's fixed buffers; the weights are re-prepared from the current parameters inside the graph on every replay."""
    key = (tuple(x.shape), x.device.index, next(self.parameters()).data_ptr())
    ent = self._eval_graphs.get(key)
    if ent is None:
        dev = x.device
        static = [x.detach().to(dev, torch.float32).contiguous().clone(), mask.detach().to(dev, torch.float32).contiguous().clone(),
                  CAM.detach().to(dev, torch.float32).contiguous().clone(), slice_ratio.detach().to(dev, torch.float64).contiguous().clone()]
        self.run_forward(*static)                       # eager warm-up: plan buffers, weight tables, kernel attributes
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, stream=E.named_stream('capture', dev), capture_error_mode='thread_local'):
                P = self.run_forward(*static)
        except RuntimeError:
            self.use_graph = False
            torch.cuda.synchronize(dev)
            return self.run_forward(x, mask, CAM, slice_ratio)
        if len(self._eval_graphs) >= 8:                 # a handful of shapes at most (bs=1 loop, batched stages)
            self._eval_graphs.clear()
        ent = self._eval_graphs[key] = (g, static, P)
    g, static, P = ent
    for dst, src in zip(static, (x, mask, CAM, slice_ratio)):
        dst.copy_(src, non_blocking=True)
    g.replay()
    return P


Generator._eval_replay = _generator_eval_replay


class _GeneratorFn(torch.autograd.Function):
    """Autograd bridge: lets user code call loss.backward() on the generator outputs; parameter gradients are
    written into .grad by the explicit backward (the anchor parameter only carries the graph edge)."""

    @staticmethod
    def forward(ctx, anchor, gen, plan, *outs):
        ctx.gen, ctx.plan = gen, plan
        plan.generation = ctx.generation = getattr(plan, 'generation', 0) + 1
        return tuple(t.clone() for t in outs)

    @staticmethod
    def backward(ctx, g_cs, g_fs, g_x1, g_x2, g_p1, g_p2):
        # gradients are ASSIGNED into .grad (the reference always zero_grad()s before backward)
        if ctx.plan.generation != ctx.generation:
            raise RuntimeError("Generator: a later forward pass of the same shape overwrote this pass's activations before its backward ran "
                               "(one activation plan per input shape); run backward before the next forward")
        S = ops.bridge_grad_scale(ctx.gen.precision)       # fp16 storage mode: scaled seeds, parameter gradients unscaled afterwards
        sc = (lambda t: None if t is None else t * S) if S != 1.0 else (lambda t: t)
        ctx.gen.run_backward(ctx.plan, sc(g_cs), sc(g_fs), sc(g_x1), sc(g_x2), sc(g_p1), sc(g_p2))
        ops.scale_inplace(ctx.gen.paramset().flat_grad, 1.0 / S)
        return (None,) * 9
