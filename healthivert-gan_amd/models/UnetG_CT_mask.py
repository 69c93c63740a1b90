"""U-Net generator with a shared encoder and two decoders (CT, mask) on the HIP path.

API mirror of the reference `models/UnetG_CT_mask.py` (define_G :63-67, DownsampleBlock :69-82, UpsampleBlock
:84-100, UnetGenerator :102-146).  The reference leaves its call commented out (models/pix2pix_model.py:96-100)
but BASELINE config #1 names it, so it is provided with identical state-dict keys
(down_blocks.N.model.*, up_blocks_ct.N.model.*, up_blocks_mask.N.model.*).  Quirk kept: use_dropout=True builds
nn.Dropout(p=True) == p=1.0, i.e. zeros in train mode (:73-78,:112-114).
"""
import torch
import torch.nn as nn
from torch.nn import init

from ._backend import engine as E
from ._backend import lib as _lib
from ._backend import ops
from ._backend import ops as _ops_
Act, rup = _ops_.Act, _ops_.rup
from .networks import init_weights  # same initialiser as the reference's private copy (:11-42)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    init_weights(net, init_type, init_gain=init_gain)
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(gpu_ids[0])
    return net


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02, gpu_ids=[]):
    return init_net(UnetGenerator(input_nc, output_nc, 5, ngf, use_dropout=use_dropout), init_type, init_gain, gpu_ids)


class DownsampleBlock(nn.Module):
    def __init__(self, in_channels, out_channels, normalize=True, dropout=0.0):
        super().__init__()
        layers = [nn.Conv2d(in_channels, out_channels, 4, stride=2, padding=1, bias=not normalize)]
        if normalize:
            layers.append(nn.BatchNorm2d(out_channels))
        layers.append(nn.LeakyReLU(0.2, inplace=True))
        if dropout > 0:
            layers.append(nn.Dropout(dropout))
        self.model = nn.Sequential(*layers)
        self.normalize, self.dropout = normalize, float(dropout)

    def forward(self, x):
        raise RuntimeError("DownsampleBlock runs inside UnetGenerator's kernel sequence")


class UpsampleBlock(nn.Module):
    def __init__(self, in_channels, out_channels, for_mask=False, dropout=0.0):
        super().__init__()
        layers = [nn.ConvTranspose2d(in_channels, out_channels, kernel_size=4, stride=2, padding=1, bias=False),
                  nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True)]
        if dropout > 0:
            layers.append(nn.Dropout(dropout))
        self.model = nn.Sequential(*layers)
        self.for_mask, self.dropout = for_mask, float(dropout)

    def forward(self, x):
        raise RuntimeError("UpsampleBlock runs inside UnetGenerator's kernel sequence")


class _UnetPlan:
    def __init__(self, net, B, H, W, device):
        self.B, self.H, self.W = B, H, W
        dt = ops.storage_dtype(net.precision)
        z = lambda h, w, C: Act(torch.zeros(B, h, w, rup(C, 4), dtype=dt, device=device), C, 0)
        self.book = E.GradBook()
        nd = len(net.down_blocks)
        P = net._pset_convs
        self.x_in = z(H, W, net.input_nc)
        self.down = []
        prev, h, w = self.x_in, H, W
        for i, blk in enumerate(net.down_blocks):
            h, w = h // 2, w // 2
            p = P['down_blocks.%d' % i]
            ent = dict(p=p, blk=blk, zbuf=z(h, w, p.cout), y=z(h, w, p.cout), stats=torch.zeros(2 * p.cout, device=device))
            ent['node'] = E.ConvNode(p, prev, ent['zbuf'] if blk.normalize else ent['y'], 2, 1, 1,
                                     'none' if blk.normalize else 'lrelu', use_bias=not blk.normalize, need_dx=i > 0)
            self.down.append(ent)
            prev = ent['y']
        self.up = {}
        for br in ('up_blocks_ct', 'up_blocks_mask'):
            ents = []
            prev = self.down[-1]['y']
            hh, ww = h, w
            for j, blk in enumerate(getattr(net, br)):
                hh, ww = hh * 2, ww * 2
                p = P['%s.%d' % (br, j)]
                last = j == nd - 1
                skip = None if last else self.down[nd - 2 - j]['y']
                cat_c = p.cout + (0 if last else skip.C)
                ent = dict(p=p, blk=blk, zbuf=z(hh, ww, p.cout), stats=torch.zeros(2 * p.cout, device=device), skip=skip)
                if last:
                    ent['out'] = torch.zeros(B, p.cout, hh, ww, device=device)
                    ent['cat'] = z(hh, ww, p.cout)
                else:
                    ent['cat'] = z(hh, ww, cat_c)
                ent['y'] = ent['cat'].slice(0, p.cout)
                ent['node'] = E.ConvNode(p, prev, ent['zbuf'], 2, 1, 1, 'none', transposed=True, use_bias=False)
                ents.append(ent)
                prev = ent['cat']
            self.up[br] = ents


class UnetGenerator(nn.Module):
    def __init__(self, input_nc, output_nc, num_downs, ngf=64, norm_layer=nn.BatchNorm2d, use_dropout=False):
        super().__init__()
        self.input_nc, self.output_nc, self.use_dropout = input_nc, output_nc, use_dropout
        self.down_blocks = nn.ModuleList()
        for i in range(num_downs):
            cin = input_nc if i == 0 else ngf * 2 ** (i - 1)
            cout = ngf * 2 ** i
            if i == num_downs - 1:
                self.down_blocks.append(DownsampleBlock(cin, cout, normalize=False, dropout=use_dropout))
            else:
                self.down_blocks.append(DownsampleBlock(cin, cout, normalize=True, dropout=use_dropout if i > 2 else 0.0))
        self.up_blocks_ct, self.up_blocks_mask = nn.ModuleList(), nn.ModuleList()
        for i in reversed(range(num_downs)):
            cin = ngf * 2 ** i if i == num_downs - 1 else ngf * 2 ** (i + 1)
            cout = ngf * 2 ** (i - 1) if i > 0 else output_nc
            self.up_blocks_ct.append(UpsampleBlock(cin, cout, for_mask=False, dropout=use_dropout if i < 3 else 0.0))
            self.up_blocks_mask.append(UpsampleBlock(cin, cout, for_mask=(i == 0), dropout=use_dropout if i < 3 else 0.0))
        self.precision = None
        self._pset = self._pset_convs = None
        self._plans = {}

    def paramset(self):
        if self._pset is None:
            convs, extra = {}, []
            for i, blk in enumerate(self.down_blocks):
                m = blk.model[0]
                convs['down_blocks.%d' % i] = E.ConvParams('down_blocks.%d' % i, m.weight, m.bias, m.in_channels, m.out_channels, 4)
                if blk.normalize:
                    extra += [blk.model[1].weight, blk.model[1].bias]
            for br in ('up_blocks_ct', 'up_blocks_mask'):
                for j, blk in enumerate(getattr(self, br)):
                    m = blk.model[0]
                    convs['%s.%d' % (br, j)] = E.ConvParams('%s.%d' % (br, j), m.weight, None, m.in_channels, m.out_channels, 4, transposed_src=True)
                    extra += [blk.model[1].weight, blk.model[1].bias]
            self._pset_convs = convs
            self._pset = E.ParamSet(convs.values(), extra)
        return self._pset

    def _plan(self, B, H, W, device):
        key = (B, H, W, str(device), ops.precision_id(self.precision))
        if key not in self._plans:
            if H % (2 ** len(self.down_blocks)) or W % (2 ** len(self.down_blocks)):
                raise NotImplementedError("UnetGenerator HIP path: H and W divisible by 2^num_downs")
            self.paramset()
            self._plans[key] = _UnetPlan(self, B, H, W, device)
        return self._plans[key]

    def _drop(self, blk, training):
        return training and self.use_dropout and blk.dropout > 0

    def run_forward(self, x, training=None):
        _lib.require_gpu(x)
        training = self.training if training is None else training
        prec = ops.precision_id(self.precision)
        B, C, H, W = x.shape
        P = self._plan(B, H, W, x.device)
        self.paramset().prep(x.device, power_iter=False)
        _lib.get().call('hv_nchw_to_nhwc', _lib.ptr(x.contiguous().float()), _lib.ptr(P.x_in.t), P.x_in.f16, B, C, H, W, P.x_in.ld, 0, _lib.stream())
        P.training = training
        for ent in P.down:
            blk = ent['blk']
            ent['node'].forward(prec)
            if blk.normalize:
                bn = blk.model[1]
                ops.norm_act_forward(ent['zbuf'], ent['y'], 'batch', training, ent['stats'], bn.weight, bn.bias, bn.running_mean,
                                     bn.running_var, bn.num_batches_tracked, act='lrelu', eps=bn.eps, momentum=bn.momentum)
            if self._drop(blk, training):
                ops.fill(ent['y'].t, 0.0)
        for br, ents in P.up.items():
            for ent in ents:
                blk, bn = ent['blk'], ent['blk'].model[1]
                ent['node'].forward(prec)
                ops.norm_act_forward(ent['zbuf'], ent['y'], 'batch', training, ent['stats'], bn.weight, bn.bias, bn.running_mean,
                                     bn.running_var, bn.num_batches_tracked, act='relu', post_sigmoid=blk.for_mask, eps=bn.eps,
                                     momentum=bn.momentum)
                if self._drop(blk, training):
                    # Dropout(p=1) zeroes the block output before the mask head's sigmoid is applied (sigmoid(0) = 0.5)
                    ops.fill(ent['y'].t if ent['skip'] is None else ent['cat'].t, 0.5 if blk.for_mask else 0.0)
                if ent['skip'] is not None:
                    ops.copy_channels(ent['skip'], ent['cat'].slice(ent['p'].cout, ent['skip'].C), mode=0)
        return P

    def outputs(self, P):
        return tuple(P.up[br][-1]['y'].nchw() for br in ('up_blocks_ct', 'up_blocks_mask'))

    def run_backward(self, P, d_ct, d_mask):
        """Gradients wrt the two outputs (B,output_nc,H,W) -> parameter .grad."""
        prec = ops.precision_id(self.precision)
        book = P.book
        book.reset()
        nd = len(self.down_blocks)
        if self.use_dropout and P.training:
            raise NotImplementedError("UnetGenerator backward with use_dropout=True (p=1.0 dropout: all gradients are zero)")
        first = True
        for br, seed in (('up_blocks_ct', d_ct), ('up_blocks_mask', d_mask)):
            ents = P.up[br]
            g_out = book.twin(ents[-1]['cat'])
            _lib.get().call('hv_nchw_to_nhwc', _lib.ptr(seed.contiguous().float()), _lib.ptr(g_out.t), g_out.f16, P.B, seed.shape[1], P.H, P.W, g_out.ld, 0, _lib.stream())
            book.mark(g_out)
            for j in range(nd - 1, -1, -1):
                ent = ents[j]
                blk, bn = ent['blk'], ent['blk'].model[1]
                gy = book.twin(ent['y'])
                gz = book.twin(ent['zbuf'])
                ops.norm_act_backward(gy, ent['y'], ent['zbuf'], gz, 'batch', P.training, ent['stats'], gamma=bn.weight, act='relu',
                                      post_sigmoid=blk.for_mask, dgamma=bn.weight.grad, dbeta=bn.bias.grad)
                if ent['skip'] is not None:   # gradient flowing into the encoder feature through the concat
                    gs = book.twin(ent['skip'])
                    ops.copy_channels(book.twin(ent['cat']).slice(ent['p'].cout, ent['skip'].C), gs, mode=0, accumulate=book.mark(gs))
                E.conv_backward(ent['node'], book, prec)
            first = False
        for i in range(nd - 1, -1, -1):
            ent = P.down[i]
            blk = ent['blk']
            if blk.normalize:
                bn = blk.model[1]
                ops.norm_act_backward(book.twin(ent['y']), ent['y'], ent['zbuf'], book.twin(ent['zbuf']), 'batch', P.training, ent['stats'],
                                      gamma=bn.weight, act='lrelu', dgamma=bn.weight.grad, dbeta=bn.bias.grad)
            E.conv_backward(ent['node'], book, prec)
        book.join()     # side-stream weight gradients
        self.paramset().finish_backward(accumulate=False)
        self.paramset().attach_grads()

    def forward(self, x):
        P = self.run_forward(x)
        ct, mk = self.outputs(P)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            anchor = next(p for p in self.parameters() if p.requires_grad)
            return _UnetFn.apply(anchor, self, P, ct, mk)
        return ct, mk


class _UnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, net, plan, ct, mk):
        ctx.net, ctx.plan = net, plan
        plan.generation = ctx.generation = getattr(plan, 'generation', 0) + 1
        return ct.clone(), mk.clone()

    @staticmethod
    def backward(ctx, g_ct, g_mk):
        if ctx.plan.generation != ctx.generation:
            raise RuntimeError("UnetGenerator: a later forward pass of the same shape overwrote this pass's activations before its backward ran")
        S = ops.bridge_grad_scale(ctx.net.precision)       # fp16 storage mode: scaled seeds, parameter gradients unscaled afterwards
        ctx.net.run_backward(ctx.plan, g_ct * S if (S != 1.0 and g_ct is not None) else g_ct, g_mk * S if (S != 1.0 and g_mk is not None) else g_mk)
        ops.scale_inplace(ctx.net.paramset().flat_grad, 1.0 / S)
        return (None,) * 5
