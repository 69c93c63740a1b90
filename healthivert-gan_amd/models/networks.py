"""PatchGAN discriminator, GAN loss, schedulers and initialisers on the HIP path.

API mirror of the parts of the reference `models/networks.py` that Pix2PixModel uses (define_D :163-206,
NLayerDiscriminator :555-602, GANLoss :212-278, get_scheduler :39-65, init_net/init_weights :68-117,
get_norm_layer :18-36).  The nn.Sequential built here is a parameter container with the reference's
state-dict keys (model.0.weight ... model.11.bias); forward/backward are explicit kernel sequences.
Generators/discriminators the hot path never instantiates (ResNet/U-Net G, pixel/seg D, WGAN-GP) are
out of scope and raise NotImplementedError.
"""
import functools
import os

import torch
import torch.nn as nn
from torch.nn import init
from torch.optim import lr_scheduler

from ._backend import engine as E
from ._backend import lib as _lib
from ._backend import ops
from ._backend import ops as _ops_
Act, rup = _ops_.Act, _ops_.rup


class Identity(nn.Module):
    def forward(self, x):
        return x


def get_norm_layer(norm_type='instance'):
    if norm_type == 'batch':
        return functools.partial(nn.BatchNorm2d, affine=True, track_running_stats=True)
    if norm_type == 'instance':
        return functools.partial(nn.InstanceNorm2d, affine=False, track_running_stats=False)
    if norm_type == 'none':
        return lambda x: Identity()
    raise NotImplementedError('normalization layer [%s] is not found' % norm_type)


def get_scheduler(optimizer, opt):
    if opt.lr_policy == 'linear':
        def lambda_rule(epoch):
            return 1.0 - max(0, epoch + opt.epoch_count - opt.n_epochs) / float(opt.n_epochs_decay + 1)
        return lr_scheduler.LambdaLR(optimizer, lr_lambda=lambda_rule)
    if opt.lr_policy == 'step':
        return lr_scheduler.StepLR(optimizer, step_size=opt.lr_decay_iters, gamma=0.1)
    if opt.lr_policy == 'plateau':
        return lr_scheduler.ReduceLROnPlateau(optimizer, mode='min', factor=0.2, threshold=0.01, patience=5)
    if opt.lr_policy == 'cosine':
        return lr_scheduler.CosineAnnealingLR(optimizer, T_max=opt.n_epochs, eta_min=0)
    return NotImplementedError('learning rate policy [%s] is not implemented', opt.lr_policy)


def init_weights(net, init_type='normal', init_gain=0.02):
    def init_func(m):
        classname = m.__class__.__name__
        if hasattr(m, 'weight') and (classname.find('Conv') != -1 or classname.find('Linear') != -1):
            if init_type == 'normal':
                init.normal_(m.weight.data, 0.0, init_gain)
            elif init_type == 'xavier':
                init.xavier_normal_(m.weight.data, gain=init_gain)
            elif init_type == 'kaiming':
                init.kaiming_normal_(m.weight.data, a=0, mode='fan_in')
            elif init_type == 'orthogonal':
                init.orthogonal_(m.weight.data, gain=init_gain)
            else:
                raise NotImplementedError('initialization method [%s] is not implemented' % init_type)
            if hasattr(m, 'bias') and m.bias is not None:
                init.constant_(m.bias.data, 0.0)
        elif classname.find('BatchNorm2d') != -1:
            init.normal_(m.weight.data, 1.0, init_gain)
            init.constant_(m.bias.data, 0.0)

    print('initialize network with %s' % init_type)
    net.apply(init_func)


def init_net(net, init_type='normal', init_gain=0.02, gpu_ids=[]):
    """The reference wraps multi-GPU nets in nn.DataParallel (:112-116); here every process drives ONE GPU and
    data parallelism is gradient all-reduce over RCCL (healthivert-gan_amd/ddp.py), so the net just moves to its device."""
    init_weights(net, init_type, init_gain=init_gain)   # on the host RNG: the same seed gives the same weights as a CPU run
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(gpu_ids[0])
    return net


def define_G(input_nc, output_nc, ngf, netG, norm='batch', use_dropout=False, init_type='normal', init_gain=0.02, gpu_ids=[]):
    if netG in ('resnet_9blocks', 'resnet_6blocks', 'unet_128', 'unet_256'):
        raise NotImplementedError("define_G('%s'): not on the HealthiVert-GAN hot path (Pix2PixModel builds inpaint_networks.Generator; "
                                  "UnetG_CT_mask.define_G is the U-Net variant provided)" % netG)
    raise NotImplementedError('Generator model name [%s] is not recognized' % netG)


def define_D(input_nc, ndf, netD, n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02, gpu_ids=[]):
    norm_layer = get_norm_layer(norm_type=norm)
    if netD == 'basic':
        net = NLayerDiscriminator(input_nc, ndf, n_layers=3, norm_layer=norm_layer)
    elif netD == 'n_layers':
        net = NLayerDiscriminator(input_nc, ndf, n_layers_D, norm_layer=norm_layer)
    elif netD in ('pixel', 'seg'):
        raise NotImplementedError("define_D('%s'): not on the HealthiVert-GAN hot path" % netD)
    else:
        raise NotImplementedError('Discriminator model name [%s] is not recognized' % netD)
    return init_net(net, init_type, init_gain, gpu_ids)


# ================================================================================================ GAN loss
class _GanLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target_is_real, mode):
        loss = torch.zeros((), device=pred.device)
        dz = torch.empty_like(pred)
        ops.gan_loss(pred.contiguous(), target_is_real, mode, loss=loss, dz=dz)
        ctx.save_for_backward(dz)
        return loss

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None, None


class GANLoss(nn.Module):
    def __init__(self, gan_mode, target_real_label=1.0, target_fake_label=0.0):
        super().__init__()
        self.register_buffer('real_label', torch.tensor(target_real_label))
        self.register_buffer('fake_label', torch.tensor(target_fake_label))
        self.gan_mode = gan_mode
        if gan_mode not in ('lsgan', 'vanilla', 'wgangp'):
            raise NotImplementedError('gan mode %s not implemented' % gan_mode)
        if target_real_label != 1.0 or target_fake_label != 0.0:
            raise NotImplementedError("GANLoss HIP path: labels 1.0 / 0.0")

    def get_target_tensor(self, prediction, target_is_real):
        return (self.real_label if target_is_real else self.fake_label).expand_as(prediction)

    def __call__(self, prediction, target_is_real):
        if self.gan_mode == 'wgangp':
            raise NotImplementedError("GANLoss('wgangp') is not on the HealthiVert-GAN hot path")
        _lib.require_gpu(prediction)
        return _GanLossFn.apply(prediction, bool(target_is_real), self.gan_mode)


# ================================================================================================ PatchGAN
CONV_STATS = os.environ.get('HV_CONV_STATS', '1') != '0'     # A/B knob: BatchNorm statistics from the producing conv's epilogue
LOSS_HEAD = os.environ.get('HV_LOSS_HEAD', '1') != '0'      # GAN loss kernel writes the logits layer's gradient carrier + bias gradient (A/B knob)
CONV_BSTATS = os.environ.get('HV_CONV_BSTATS', '1') != '0'   # A/B knob: BatchNorm backward sums from the epilogue of the data gradient that writes dy
FUSE_NORM_ACT = os.environ.get('HV_FUSE_NORM_ACT', '1') != '0'     # A/B knob, see _DiscPlan-based run_backward
HEAD_NORM = os.environ.get('HV_HEAD_NORM', '1') != '0'     # A/B knob (same bits): the last normalisation + LeakyReLU made where the logits layer stages its input


class _DiscPlan:
    def __init__(self, net, B, H, W, device):
        self.B, self.H, self.W = B, H, W
        dt = ops.storage_dtype(net.precision)        # fp16 buffers in the fp16 mode; input image, logits and dx stay fp32
        z = lambda h, w, C: Act(torch.zeros(B, h, w, rup(C, 4), dtype=dt, device=device), C, 0)
        self.book = E.GradBook()
        self.x4 = z(H, W, 1)
        self.layers = []
        h, w = H, W
        prev = None
        for li, L in enumerate(net._spec):
            ho, wo = ops.conv_out_size(h, 4, L['stride'], 1, 1), ops.conv_out_size(w, 4, L['stride'], 1, 1)
            p = net._pset_convs[li]
            ent = dict(spec=L, p=p)
            if li == 0:
                ent['y'] = z(ho, wo, p.cout)
                ent['node'] = None     # built per call (input tensor changes)
            elif L['last']:
                self.logits = torch.zeros(B, 1, ho, wo, dtype=torch.float32, device=device)
                ent['y'] = Act(self.logits.view(B, ho, wo, 1))
                ent['node'] = E.ConvNode(p, prev, ent['y'], L['stride'], 1, 1, 'none', use_bias=True)
                self.g_logits = z(ho, wo, 1)
            else:
                ent['z'] = z(ho, wo, p.cout)
                ent['y'] = z(ho, wo, p.cout)
                ent['stats'] = torch.zeros(2 * B * p.cout, dtype=torch.float32, device=device)
                ent['node'] = E.ConvNode(p, prev, ent['z'], L['stride'], 1, 1, 'none', use_bias=L['bias'])
            prev = ent['y']
            h, w = ho, wo
            self.layers.append(ent)
        self.dx = torch.zeros(B, 1, H, W, dtype=torch.float32, device=device)
        self.pending = None        # weakref to the autograd token of the nn.Module-API forward that owns these activations


def _stat_momentum(m, order):
    if order == 'swapped_first':
        return m / (1.0 - m * (1.0 - m))
    if order == 'swapped_second':
        return m * (1.0 - m)
    return m


class NLayerDiscriminator(nn.Module):
    """PatchGAN: Conv(k4,s2)+LReLU, (n_layers-1)x[Conv(k4,s2)+Norm+LReLU], Conv(k4,s1)+Norm+LReLU, Conv(k4,s1)->1."""

    def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=nn.BatchNorm2d):
        super().__init__()
        if type(norm_layer) == functools.partial:
            use_bias = norm_layer.func == nn.InstanceNorm2d
            norm_cls = norm_layer.func
        else:
            use_bias = norm_layer == nn.InstanceNorm2d
            norm_cls = norm_layer
        print(norm_layer)
        if input_nc != 1:
            raise NotImplementedError("NLayerDiscriminator HIP path: input_nc == 1")
        if norm_cls not in (nn.BatchNorm2d, nn.InstanceNorm2d):
            raise NotImplementedError("NLayerDiscriminator HIP path: norm in {batch, instance}")
        self.norm_kind = 'batch' if norm_cls == nn.BatchNorm2d else 'instance'
        kw, padw = 4, 1
        seq = [nn.Conv2d(input_nc, ndf, kernel_size=kw, stride=2, padding=padw), nn.LeakyReLU(0.2, True)]
        spec = [dict(conv=0, norm=None, stride=2, bias=True, last=False)]
        nf_mult = 1
        for n in range(1, n_layers):
            nf_prev, nf_mult = nf_mult, min(2 ** n, 8)
            spec.append(dict(conv=len(seq), norm=len(seq) + 1, stride=2, bias=use_bias, last=False))
            seq += [nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=2, padding=padw, bias=use_bias),
                    norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        nf_prev, nf_mult = nf_mult, min(2 ** n_layers, 8)
        spec.append(dict(conv=len(seq), norm=len(seq) + 1, stride=1, bias=use_bias, last=False))
        seq += [nn.Conv2d(ndf * nf_prev, ndf * nf_mult, kernel_size=kw, stride=1, padding=padw, bias=use_bias),
                norm_layer(ndf * nf_mult), nn.LeakyReLU(0.2, True)]
        spec.append(dict(conv=len(seq), norm=None, stride=1, bias=True, last=True))
        seq += [nn.Conv2d(ndf * nf_mult, 1, kernel_size=kw, stride=1, padding=padw)]
        self.model = nn.Sequential(*seq)
        self._spec = spec
        self.precision = None
        self._pset = None
        self._pset_convs = None
        self._plans = {}

    def paramset(self):
        if self._pset is None:
            convs, extra = [], []
            for li, L in enumerate(self._spec):
                m = self.model[L['conv']]
                cin, cout = m.in_channels, m.out_channels
                if li == 0:
                    convs.append(E.ConvParams('model.%d' % L['conv'], m.weight, m.bias, cin, cout, 4, cin_fwd=1, cin_wg=4))
                else:
                    convs.append(E.ConvParams('model.%d' % L['conv'], m.weight, m.bias, cin, cout, 4))
                if L['norm'] is not None and self.norm_kind == 'batch':
                    nm = self.model[L['norm']]
                    extra += [nm.weight, nm.bias]
            self._pset_convs = convs
            self._pset = E.ParamSet(convs, extra)
        return self._pset

    def load_state_dict(self, *args, **kwargs):
        r = super().load_state_dict(*args, **kwargs)
        if self._pset is not None:
            self._pset.weights_changed()      # (copied into the same storage: the prepared weight tables are stale)
        return r

    def _plan(self, B, H, W, device, slot=0):
        """slot 0 is the explicit executor's plan (Pix2PixModel's fused step); the nn.Module API takes one plan per forward whose
        autograd graph is still alive (slot 1, 2, ...): `pred_fake = D(fake); pred_real = D(real); loss.backward()` -- the
        reference's own backward_D (pix2pix_model.py:267-283) -- keeps both passes' activations until their backward ran."""
        key = (B, H, W, str(device), slot, ops.precision_id(self.precision))
        if key not in self._plans:
            self.paramset()
            self._plans[key] = _DiscPlan(self, B, H, W, device)
        return self._plans[key]

    def _free_plan_slot(self, B, H, W, device):
        slot = 1
        while True:
            P = self._plans.get((B, H, W, str(device), slot, ops.precision_id(self.precision)))
            if P is None or P.pending is None or P.pending() is None:
                return slot
            slot += 1

    # ---------------------------------------------------------------- explicit forward / backward
    def run_forward(self, x, training=None, prep=True, groups=1, stat_order=None, slot=0):
        """x: (B,1,H,W) device tensor -> plan; logits in plan.logits (B,1,Ho,Wo).  BatchNorm running statistics are
        updated when training (every call, like the reference's three calls per step).  groups=2 treats the two halves of
        the batch as two consecutive calls (separate batch statistics, running stats updated half by half): the fake and
        the real pass of one discriminator update in a single launch sequence.
        stat_order: the train step runs the REAL pass before the FAKE pass (so that it overlaps the generator forward) while the
        reference updates the running statistics fake-then-real; 'swapped_first' / 'swapped_second' use momenta m/(1-m(1-m)) and
        m(1-m), which leave exactly the reference's running statistics after the pair:
        (1-m)^2 r + m(1-m) fake + m real."""
        _lib.require_gpu(x)
        training = self.training if training is None else training
        prec = ops.precision_id(self.precision)
        x = x.contiguous().float()
        B, _, H, W = x.shape
        P = self._plan(B, H, W, x.device, slot)
        P.book.join()   # weight gradients of the previous backward still read this plan's activations on the side stream
        if prep:      # prep='if_stale': skipped when the tables in memory were written from the current weights (the train step: engine.ParamSet.prep)
            self.paramset().prep(x.device, power_iter=False, only_if_stale=(prep == 'if_stale'))
        xin = Act(x.view(B, H, W, 1))
        P.x_in = xin
        head_xn = None
        for li, ent in enumerate(P.layers):
            L = ent['spec']
            if li == 0:
                ent['node'] = E.ConvNode(ent['p'], xin, ent['y'], L['stride'], 1, 1, 'lrelu', use_bias=True)
                ent['node'].forward(prec)
                continue
            if L['last']:
                ent['node'].forward(prec, xn=head_xn, x_raw=P.layers[li - 1]['z'] if head_xn is not None else None)
                break
            nm = self.model[L['norm']]
            # the layer below the logits: its normalisation + LeakyReLU is applied by the logits layer's kernel where it stages its input (hv_conv_desc.xn_*; that
            # kernel also stores the normalised map the backward reads) -- the normalisation call below then only finalises the statistics.  Asked of the C
            # dispatch once per plan and mode (HV_HEAD_NORM=0: always the separate pass)
            head_xn = None
            if HEAD_NORM and P.layers[li + 1]['spec']['last'] and ent['z'].f16:
                cand = (ent['stats'], nm.weight if self.norm_kind == 'batch' else None, nm.bias if self.norm_kind == 'batch' else None,
                        groups if self.norm_kind == 'batch' else B, 'lrelu', ent['y'])
                key = ('head_xn', self.norm_kind, groups, prec)
                if key not in ent:
                    ent[key] = bool(P.layers[li + 1]['node'].forward(prec, xn=cand, probe=True, x_raw=ent['z']))
                if ent[key]:
                    head_xn = cand
            # BatchNorm statistics out of the conv's own epilogue where its kernel has one (the 4x4 stride-2 layers): the normalisation then
            # skips its reduction pass over z (HV_CONV_STATS=0: always reduce)
            parts = 0
            if CONV_STATS and self.norm_kind == 'batch' and training:      # (groups > 1: the partials are per image tile, in image order: the finalize sums each group's share)
                if 'parts' not in ent:
                    ent['parts'] = int(ent['node'].stats_parts(prec))
                    ent['partials'] = torch.zeros(max(1, ent['parts']) * ent['p'].cout * 2, dtype=torch.float32, device=x.device)
                parts = ent['parts']
            ent['node'].forward(prec, stats=ent['partials'] if parts else None)
            y_out = None if head_xn is not None else ent['y']
            if self.norm_kind == 'batch':
                ops.norm_act_forward(ent['z'], y_out, 'batch', training, ent['stats'], nm.weight, nm.bias, nm.running_mean,
                                     nm.running_var, nm.num_batches_tracked, act='lrelu', eps=nm.eps, momentum=_stat_momentum(nm.momentum, stat_order),
                                     groups=groups, partials=ent['partials'] if parts else None, n_partials=parts)
            else:
                ops.norm_act_forward(ent['z'], y_out, 'instance', training, ent['stats'], act='lrelu', eps=nm.eps)
        P.training, P.groups = training, groups
        return P

    def loss_backward(self, P, target_is_real, mode, loss, grad_weight, need_dx=False, param_grads=True, accumulate=False, loss_weight=1.0, dz=None):
        """GAN loss on P.logits + backward.  fp16 storage mode: the loss kernel writes d loss / d logit straight into the logits layer's gradient carrier
        and sums its bias gradient (hv_gan_loss_head) -- the copy, the column-sum pass and its finalize leave the chain between forward and backward."""
        last = P.layers[-1]
        if LOSS_HEAD and P.g_logits.f16 and P.g_logits.t.shape[-1] == 4 and P.g_logits.coff == 0:
            pl = last['p']
            want_db = param_grads and pl.bias is not None and last['node'].use_bias
            if not ops.gan_loss_pair(P.logits, target_is_real, loss, Act(P.g_logits.t, 4, 0), mode=mode, loss_weight=loss_weight, grad_weight=grad_weight,
                                     dbias=pl.bias.grad if want_db else None, dbias_accumulate=accumulate):
                ops.gan_loss(P.logits, target_is_real, mode, loss=loss, loss_weight=loss_weight, grad_weight=grad_weight, carrier=Act(P.g_logits.t, 4, 0),
                             dbias=pl.bias.grad if want_db else None, dbias_accumulate=accumulate)
            return self.run_backward(P, None, need_dx=need_dx, param_grads=param_grads, accumulate=accumulate, logits_ready=True)
        if dz is None:
            dz = torch.empty_like(P.logits)
        ops.gan_loss(P.logits, target_is_real, mode, loss=loss, loss_weight=loss_weight, dz=dz, grad_weight=grad_weight)
        return self.run_backward(P, dz, need_dx=need_dx, param_grads=param_grads, accumulate=accumulate)

    def loss_backward_halves(self, P, mode, loss_fake, loss_real, grad_weight, dz=None):
        """The batched fake | real pass (run_forward(..., groups=2) on [fake; real]): GAN loss of each half against its own target + ONE backward.
        fp16 storage mode: each half's loss kernel writes its part of the logits layer's gradient carrier and adds its bias gradient
        (hv_gan_loss_head), as loss_backward does for a whole batch."""
        last = P.layers[-1]
        B = P.B // 2
        if LOSS_HEAD and P.g_logits.f16 and P.g_logits.t.shape[-1] == 4 and P.g_logits.coff == 0:
            pl = last['p']
            want_db = pl.bias is not None and last['node'].use_bias
            # both halves in one single-workgroup launch where they fit (four tiny dependent launches between forward and backward -> one)
            if not ops.gan_loss_pair(P.logits[:B], False, loss_fake, Act(P.g_logits.t[:B], 4, 0), P.logits[B:], True, loss_real, Act(P.g_logits.t[B:], 4, 0),
                                     mode=mode, grad_weight=grad_weight, dbias=pl.bias.grad if want_db else None):
                for half, (real, loss) in enumerate(((False, loss_fake), (True, loss_real))):
                    sl = slice(half * B, (half + 1) * B)
                    ops.gan_loss(P.logits[sl], real, mode, loss=loss, grad_weight=grad_weight, carrier=Act(P.g_logits.t[sl], 4, 0),
                                 dbias=pl.bias.grad if want_db else None, dbias_accumulate=bool(half))
            return self.run_backward(P, None, need_dx=False, param_grads=True, accumulate=False, logits_ready=True)
        if dz is None:
            dz = torch.empty_like(P.logits)
        ops.gan_loss(P.logits[:B], False, mode, loss=loss_fake, dz=dz[:B], grad_weight=grad_weight)
        ops.gan_loss(P.logits[B:], True, mode, loss=loss_real, dz=dz[B:], grad_weight=grad_weight)
        return self.run_backward(P, dz, need_dx=False, param_grads=True, accumulate=False)

    def run_backward(self, P, dlogits, need_dx=False, param_grads=True, accumulate=False, logits_ready=False):
        """dlogits: (B,1,Ho,Wo) gradient of the loss wrt the logits.  Fills kernel-layout weight gradients and the
        bias / affine .grad (accumulating when `accumulate`); call finish() afterwards.  Returns d loss / d input."""
        prec = ops.precision_id(self.precision)
        book = P.book
        book.reset()
        B = P.B
        last = P.layers[-1]
        if not logits_ready:      # (loss_backward: the loss kernel already wrote the carrier and the logits layer's bias gradient)
            ops.copy_channels(Act(dlogits.contiguous().view(B, last['y'].H, last['y'].W, 1)), P.g_logits, mode=0)
        book.twins[id(last['y'].t)] = P.g_logits.t
        # the input view changes every call: its gradient always lives in P.dx
        book.twins.pop(getattr(P, '_in_id', None), None)
        P._in_id = id(P.x_in.t)
        book.twins[P._in_id] = P.dx.view(B, P.H, P.W, 1)
        fuse0 = E.FUSE_ACT and len(P.layers) > 1 and P.layers[0]['node'].act != 'none' and not P.layers[1]['node'].shift
        # a normalised layer's LeakyReLU' rides in the data-gradient epilogue of the layer that consumes its output z (one consumer), so the
        # normalisation's backward starts from the gradient at ITS output and never reads z (HV_FUSE_NORM_ACT=0: the norm kernels apply it)
        fuse_n = E.FUSE_ACT and FUSE_NORM_ACT
        for li in range(len(P.layers) - 1, -1, -1):
            ent = P.layers[li]
            L, node = ent['spec'], ent['node']
            prev_normed = li >= 2 and not P.layers[li - 1]['spec']['last'] and not node.shift       # layer li - 1 has a norm + LeakyReLU
            if li == 0:
                x4 = None
                if param_grads:
                    ops.copy_channels(P.x_in, P.x4, mode=0)
                node.need_dx = need_dx
                E.conv_backward(node, book, prec, dbias_accumulate=accumulate, wgrad_accumulate=accumulate, wgrad=param_grads,
                                x_wg=P.x4 if param_grads else None, premultiplied=fuse0)
                break
            if not L['last']:
                nm = self.model[L['norm']]
                gy, gz = book.twin(ent['y']), book.twin(ent['z'])
                bn = self.norm_kind == 'batch'
                bparts = ent.get('bparts_used', 0)      # the consumer's data gradient (layer li + 1, a moment ago) summed for this normalisation
                ops.norm_act_backward(gy, ent['y'], ent['z'], gz, self.norm_kind, P.training, ent['stats'],
                                      gamma=nm.weight if bn else None, act='none' if fuse_n else 'lrelu',
                                      dgamma=nm.weight.grad if (bn and param_grads) else None,
                                      dbeta=nm.bias.grad if (bn and param_grads) else None, param_accumulate=accumulate,
                                      groups=P.groups, partials=ent['bpartials'] if bparts else None, n_partials=bparts)
            # the stem's output has one consumer (layer 1): its LeakyReLU' rides in layer 1's data-gradient epilogue
            mul_x = P.layers[0]['node'].act if (li == 1 and fuse0) else ('lrelu' if (prev_normed and fuse_n) else None)
            # ... and where layer li - 1 is batch-normalised, the sums of ITS backward (sum g, sum g * xhat over the gradient this launch writes) leave
            # this data gradient's epilogue: the normalisation's backward then skips its reduction pass over g and z (HV_CONV_BSTATS=0: it reduces)
            bn_arg = None
            if prev_normed and fuse_n and mul_x and CONV_BSTATS and self.norm_kind == 'batch' and P.training:
                pe = P.layers[li - 1]
                bp = pe.setdefault('bparts', {})      # by statistics groups: a plan serves the batched fake | real pass (two groups) and a plain pass of the same size
                if P.groups not in bp:
                    bp[P.groups] = self._bstats_parts(node, pe, P, book, prec, mul_x)
                    # (a buffer per group count: captured graphs of both uses keep their own addresses)
                    pe.setdefault('bpartials_by', {})[P.groups] = torch.zeros(max(1, bp[P.groups]) * pe['p'].cout * 2, dtype=torch.float32, device=pe['z'].t.device)
                if bp[P.groups]:
                    pe['bpartials'] = pe['bpartials_by'][P.groups]
                    bn_arg = (pe['z'], pe['stats'], P.groups, pe['bpartials'])
                pe['bparts_used'] = bp[P.groups] if bn_arg else 0
            elif prev_normed:
                P.layers[li - 1]['bparts_used'] = 0
            E.conv_backward(node, book, prec, dbias_accumulate=accumulate, wgrad_accumulate=accumulate, wgrad=param_grads, dbias_done=bool(logits_ready and L['last']),
                            mul_x=mul_x, bn=bn_arg)
        if need_dx:
            g = book.twin(P.x_in)
            return g.t.view(B, 1, P.H, P.W)
        return None

    def _bstats_parts(self, node, pe, P, book, prec, mul_x):
        """Parts of the batch-norm backward sums that node's data gradient (exactly as conv_backward issues it) would write for the normalised layer
        pe below it; 0 = its kernel has no such epilogue."""
        p = node.p
        gy = book.twin(node.y)
        gfull = Act(gy.t, p.coutP, gy.coff)
        gx = book.twin(node.x)
        gx = Act(gx.t, node.dx_c or p.cin_fwd, gx.coff)
        if (id(gx.t), gx.coff, gx.C) in book.written:       # (an accumulating data gradient is not a whole sum; never the case in this chain)
            return 0
        return int(ops.conv2d_bstats_parts(gfull, p.w_bwd, gx, node.k, node.s, node.pad, node.d, transposed=True, accumulate=0, w_h=p.w_bwd_h, w_t=p.w_bwd_t,
                                           precision=prec, mul=(Act(node.x.t, p.cin_fwd, node.x.coff), mul_x), bn=(pe['z'], pe['stats'], P.groups, None)))

    def finish(self):
        for P in self._plans.values():
            P.book.join()
        self.paramset().finish_backward(accumulate=False)
        self.paramset().attach_grads()

    # ---------------------------------------------------------------- nn.Module API
    def forward(self, input):
        if torch.is_grad_enabled() and (input.requires_grad or any(p.requires_grad for p in self.parameters())):
            B, _, H, W = input.shape
            P = self.run_forward(input, slot=self._free_plan_slot(B, H, W, input.device))
            anchor = next((p for p in self.parameters() if p.requires_grad), None)   # carries the graph edge when only the weights need gradients
            return _DiscFn.apply(input, anchor, self, P, P.logits)
        return self.run_forward(input).logits.clone()


class _PlanToken:
    """Alive as long as the autograd node that owns a plan's activations is."""


def grads_are_fresh(net):
    """True when nothing has been written into the net's gradients since the last zero_grad(): the next backward ASSIGNS, later
    ones accumulate -- torch.autograd's .grad semantics for the nn.Module API of the explicit-backward networks.  FusedAdam.zero_grad
    flags the first parameter; torch optimisers either drop .grad (set_to_none) or zero it (then accumulating is right anyway)."""
    ps = list(net.parameters())
    fresh = getattr(ps[0], '_hv_fresh', False) or ps[0].grad is None
    for p in ps:
        p._hv_fresh = False
    return fresh


class _DiscFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, anchor, net, plan, logits):
        import weakref
        ctx.net, ctx.plan, ctx.need_dx = net, plan, x.requires_grad
        ctx.token = _PlanToken()
        plan.pending = weakref.ref(ctx.token)
        return logits.clone()

    @staticmethod
    def backward(ctx, g):
        net, plan = ctx.net, ctx.plan
        if plan.pending is None or plan.pending() is not ctx.token:
            raise RuntimeError("NLayerDiscriminator: the activations of this forward pass were overwritten before its backward ran")
        pg = any(p.requires_grad for p in net.parameters())
        acc = pg and not grads_are_fresh(net)
        if pg:
            net.paramset().attach_grads()
        # fp16 storage mode: scaled seeds (ops.bridge_grad_scale); gradients already in .grad are scaled up first when this call accumulates
        S = ops.bridge_grad_scale(net.precision)
        flat = net.paramset().flat_grad if pg else None
        if acc:
            ops.scale_inplace(flat, S)
        dx = net.run_backward(plan, g.contiguous() * S if S != 1.0 else g.contiguous(), need_dx=ctx.need_dx, param_grads=pg, accumulate=acc)
        if pg:
            net.finish()
            ops.scale_inplace(flat, 1.0 / S)
        plan.pending = None
        if dx is not None:
            dx = dx.clone()
            ops.scale_inplace(dx, 1.0 / S)
        return dx, None, None, None, None
