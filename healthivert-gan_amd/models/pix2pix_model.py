"""Pix2PixModel on the HIP path: the 2.5D coarse-to-fine generator + three PatchGAN discriminators train step.

API mirror of the reference `models/pix2pix_model.py` (options :51-72, construction :74-135, set_input :137-175,
forward :180-264, backward_D_1/2/3 :267-314, backward_G :317-354, optimize_parameters :356-382).  Same attribute,
loss and visual names (train.py:56-99 reads them), but the step is an explicit sequence of libhvgan kernels:
no autograd tape, no per-sample Python loops, no `.item()` host syncs (the SHRM row bounds are computed on the
device), fused multi-tensor Adam.  Losses stay on the device until get_current_losses() converts them.
"""
import ctypes

import torch

from ._backend import ddp
from ._backend import engine
from ._backend import lib as _lib
from ._backend import ops
from ._backend import lib as _lib_
ptr, stream = _lib_.ptr, _lib_.stream
from ._backend import optim as _optim_
FusedAdam = _optim_.FusedAdam
from . import networks
from .base_model import BaseModel
from .edge_operator import Sobel
from .inpaint_networks import Generator




def _os_environ_graph():
    import os
    return os.environ.get('HV_GRAPH', '1') != '0'


def diceCoeff(pred, gt, eps=1e-5, activation='sigmoid'):
    """Dice coefficient (reference :13-39); host-side helper kept for API parity (the train step uses the fused
    hv_generator_losses kernel instead)."""
    if activation not in (None, 'none', 'sigmoid', 'softmax2d'):
        raise NotImplementedError("Activation implemented for sigmoid and softmax2d")
    if activation == 'sigmoid':
        pred = torch.sigmoid(pred)
    elif activation == 'softmax2d':
        pred = torch.softmax(pred, dim=1)
    n = gt.shape[0]
    p, g = pred.reshape(n, -1), gt.reshape(n, -1)
    return ((2 * (g * p).sum(1) + eps) / (p.sum(1) + g.sum(1) + eps)).sum() / n


class Pix2PixModel(BaseModel):
    @staticmethod
    def modify_commandline_options(parser, is_train=True):
        parser.set_defaults(norm='batch', netG='unet_256', dataset_mode='aligned')
        if is_train:
            parser.set_defaults(pool_size=0, gan_mode='vanilla')
            parser.add_argument('--lambda_L1', type=float, default=200.0, help='weight for L1 loss')
        return parser

    def __init__(self, opt):
        # launched by `python -m torch.distributed.run --nproc-per-node N train.py ...`: join the job and drive THIS rank's GPU
        # (train.py itself stays unedited; it passes --gpu_ids 0 to every rank)
        local = ddp.init_from_env()
        if local is not None and opt.gpu_ids:
            opt.gpu_ids = [local]
            torch.cuda.set_device(local)
        BaseModel.__init__(self, opt)
        if not self.gpu_ids:
            raise RuntimeError("Pix2PixModel (healthivert-gan_amd) needs an MI355X: set --gpu_ids 0; there is no CPU path")
        _lib.get()   # fail loudly if libhvgan.so is missing
        self.loss_names = ['G_GAN', 'G_maskL1', 'G_Dice', 'coarse_Dice', 'edge', 'D_real_1', 'D_fake_1', 'D_real_2', 'D_fake_2',
                           'D_real_3', 'D_fake_3', 'h']
        self.visual_names = ['real_A', 'fake_B', 'fake_B_mask_raw', 'normal_vert', 'coarse_seg_binary', 'fake_B_coarse', 'real_B',
                             'mask', 'fake_B_raw', 'real_B_mask', 'CAM', 'real_edges', 'fake_B_local']
        self.model_names = ['G', 'D_1', 'D_2', 'D_3'] if self.isTrain else ['G']
        self.netG = Generator({'input_dim': 1, 'ngf': 16}, True)
        self.netG.to(self.device)
        self.sobel_edge = Sobel(requires_grad=False).to(self.device)
        self.half_band = 35
        if self.isTrain:
            for k in (1, 2, 3):
                setattr(self, 'netD_%d' % k, networks.define_D(opt.input_nc, opt.ndf, opt.netD, opt.n_layers_D, opt.norm,
                                                               opt.init_type, opt.init_gain, self.gpu_ids))
            self.criterionGAN = networks.GANLoss(opt.gan_mode).to(self.device)
            self.criterionL1 = torch.nn.L1Loss()
            if opt.gan_mode not in ('vanilla', 'lsgan'):
                raise NotImplementedError("Pix2PixModel HIP path: gan_mode in {vanilla, lsgan}")
            self.optimizer_G = FusedAdam(self.netG.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizer_D_1 = FusedAdam(self.netD_1.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizer_D_2 = FusedAdam(self.netD_2.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizer_D_3 = FusedAdam(self.netD_3.parameters(), lr=opt.lr, betas=(opt.beta1, 0.999))
            self.optimizers += [self.optimizer_G, self.optimizer_D_1, self.optimizer_D_2, self.optimizer_D_3]
        self._loss_buf = torch.zeros(32, dtype=torch.float32, device=self.device)
        self._bufs = {}
        self._in = {}
        self._shapes, self._cur = {}, None      # per batch shape: input buffers, warm-up count, captured graphs
        self._graphs = None
        self._inline_exchange = False
        self._eager_steps = 0
        self.use_graph = _os_environ_graph()
        self.grad_sync = ddp.GradSync() if self.isTrain else None
        if self.isTrain:       # every rank starts from rank 0's initial weights (a no-op without a process group)
            ddp.broadcast_parameters([self.netG, self.netD_1, self.netD_2, self.netD_3])
        import os as _os
        # gradient (loss) scale of the fp16 storage mode: activation gradients of this model are ~1e-5 .. 1e-7, i.e. subnormal or zero in
        # fp16.  Every gradient seed of the step is multiplied by a power of two S (exact), the whole backward is linear in its seeds, and the
        # flat parameter-gradient buffer of each network is multiplied by 1/S (exact) before anything reads it.  S = 1 in the fp32 mode.
        fp16 = ops.default_precision() == _lib.F16
        self.grad_scale = float(_os.environ.get('HV_GRAD_SCALE', '8192' if fp16 else '1'))
        if self.grad_scale <= 0 or (self.grad_scale != 1 and not float(self.grad_scale).is_integer()) or int(self.grad_scale) & (int(self.grad_scale) - 1):
            raise ValueError('HV_GRAD_SCALE must be a power of two')
        self.concurrent_d = _os.environ.get('HV_CONCURRENT_D', '1') != '0'
        # fake | real discriminator passes as ONE 2B-sample launch sequence (per-half BatchNorm groups).  Round 2: no gain beside the three-stream overlap;
        # re-measured at the end of round 3 with the pipelined 4x4 kernels (one round of one workgroup per CU at bs 16): 8.18 -> 8.09 ms in three same-box
        # pairs, although it gives up the real-image passes' overlap with the generator forward.  Both the single-process and the data-parallel step take it.
        self.batch_d = _os.environ.get('HV_BATCH_D', '1') != '0'
        # the three phases are captured as ONE graph (7.81 -> 7.70 ms over four same-box pairs: two graph-launch boundaries less) -- except under the cut
        # data-parallel schedule, which issues its collectives between the graphs
        # data-parallel step schedule (one process per GPU); every collective of the step goes to one communicator on one stream in the order D_1, D_2, D_3, G:
        #   'captured' (RCCL only): the step as ONE hipGraph with the two collectives inside it, both on the main branch (ncclAllReduce, ncclAvg, ddp.RcclComm): the
        #       three discriminators' gradients -- one arena -- where their streams join, the generator's between its backward and its Adam step.  No graph
        #       cut, no host in the loop, no edge between branches, the order D, G on every rank by construction.
        #   'graphs': the step cut into its three graphs where the exchanges belong, the same collectives issued eagerly between them on the exchange stream
        #       (the main stream waits for each).  The fallback for runtimes that refuse to capture RCCL kernels, and the only schedule for gloo.
        #   'overlapped' (RCCL only): 'captured' with one collective per discriminator, issued the moment D_k's gradients are final and chained D_1 -> D_2 ->
        #       D_3 -> G by events (ddp.GradSync.reduce_branch): D_k's mean runs beside the other discriminators' passes, at the price of edges between the
        #       graph's branches (this runtime executes branch-crossing edges poorly: +0.3 ms in a one-rank group where 'captured' costs nothing).
        #   'auto' (default): gloo -> 'graphs'; RCCL -> dp_preflight() runs ALL on the job's first batch, checks that every rank ends with the same
        #       weights, keeps the fastest correct one and puts the weights back (the preflight steps are not training steps).
        self.dp_schedule = _os.environ.get('HV_DP_SCHEDULE', 'auto')
        if self.dp_schedule == 'phases':      # (the twelve-phase schedule of rounds 1-3 is gone; old launch scripts keep working)
            import warnings
            warnings.warn("HV_DP_SCHEDULE=phases is deprecated: taking 'graphs'", DeprecationWarning)
            self.dp_schedule = 'graphs'
        if self.dp_schedule not in ('auto', 'captured', 'overlapped', 'graphs'):
            raise ValueError("HV_DP_SCHEDULE must be 'auto', 'captured', 'overlapped' or 'graphs'")
        self.dp_preflight_record = None
        self._in_preflight = False
        self.real_first = _os.environ.get('HV_REAL_FIRST', '1') != '0'   # D real passes overlap the generator forward
        self._through_ab = False

    # tensors forward()/backward bind as attributes; they live in per-shape buffers, so the names follow the active batch shape
    _STEP_OUTPUTS = ('fake_B', 'fake_B_coarse', 'fake_B_local', 'real_B_local', 'fake_B_mask_raw', 'coarse_seg_binary', 'coarse_seg_sigmoid',
                     'fake_B_mask_sigmoid', 'x_stage1', 'fake_B_raw', 'pred1_h', 'pred2_h', 'real_edges', 'fake_edges', '_gplan', '_rows', '_dxs')

    # discriminator k's batched input [fake_k | real_k] (2B samples): the step's producers write straight into its halves
    _PAIRED = {'real_B': 1, 'real_B_mask': 2}

    def _pair_buffer(self, k, shape):
        """D_k's 2B-sample input buffer for the current batch shape (lives with the inputs: its upper half IS an input)."""
        key = 'dcat%d' % k
        b = self._in.get(key)
        if b is None or b.shape[1:] != tuple(shape[1:]) or b.shape[0] != 2 * shape[0]:
            b = self._in[key] = torch.zeros((2 * shape[0],) + tuple(shape[1:]), dtype=torch.float32, device=self.device)
        return b

    # ---------------------------------------------------------------- inputs
    def set_input(self, input):
        """Unpack a batch dict (reference models/pix2pix_model.py:137-175).  The tensors land in persistent device buffers, one set
        per batch shape: the captured step graphs read their inputs from fixed addresses, and a shape that comes back (the partial
        last batch of every epoch, then full batches again) finds its buffers, its warm-up count and its captured graphs again
        instead of re-capturing.  The attributes (and get_current_visuals()) are views of these buffers: the next set_input of the
        same shape overwrites them -- clone what must outlive a step."""
        AtoB = self.opt.direction == 'AtoB'
        key = tuple(input['A_mask'].shape)
        st = self._shapes.get(key)
        if st is None:
            st = self._shapes[key] = {'in': {}, 'graphs': None, 'eager': 0}
        if st is not self._cur:
            if self._cur is not None:
                self._cur.update(graphs=self._graphs, eager=self._eager_steps,
                                 outs={n: getattr(self, n) for n in self._STEP_OUTPUTS if hasattr(self, n)})
            self._cur, self._in = st, st['in']
            self._graphs, self._eager_steps = st['graphs'], st['eager']
            for n, v in st.get('outs', {}).items():     # a graph replay does not re-run the Python that binds these names
                setattr(self, n, v)

        def put(name, t, dtype):
            b = self._in.get(name)
            if b is None or b.shape != t.shape or b.dtype != dtype:
                pair = self._PAIRED.get(name) if (self.isTrain and self.batch_d) else None
                if pair is not None:      # the real image of D_k is the upper half of D_k's 2B-sample input buffer (fake | real): no copy into it per step
                    b = self._pair_buffer(pair, t.shape)[t.shape[0]:]
                else:
                    b = torch.empty(t.shape, dtype=dtype, device=self.device)
                self._in[name] = b
                self._graphs = None        # input addresses changed: captured graphs are stale,
                self._eager_steps = 0      # and the new shape needs its own eager warm-up (plans, tables) before a capture
            b.copy_(t, non_blocking=True)
            return b
        self.real_B = put('real_B', input['B' if AtoB else 'A'], torch.float32)
        self.real_B_mask = put('real_B_mask', input['A_mask'], torch.float32)
        self.real_A = put('real_A', input['A' if AtoB else 'B'], torch.float32)
        self.CAM = put('CAM', input['CAM'], torch.float32)
        self.normal_vert = put('normal_vert', input['normal_vert'], torch.float32)
        self.mask = put('mask', input['mask'], torch.float32)
        self.height = put('height', input['height'], torch.int64)
        self.slice_ratio = put('slice_ratio', input['slice_ratio'], torch.float64)
        self.x1 = put('x1', input['x1'], torch.int64)
        self.x2 = put('x2', input['x2'], torch.int64)
        self.maxheight = put('maxheight', input['h2'], torch.int64)
        self.image_paths = input['A_paths' if AtoB else 'B_paths']

    def _buf(self, name, like=None, shape=None, dtype=torch.float32):
        key = (name, tuple(like.shape) if like is not None else tuple(shape))
        b = self._bufs.get(key)
        if b is None:
            b = torch.zeros(key[1], dtype=dtype, device=self.device)
            self._bufs[key] = b
        return b

    @property
    def offset_flow(self):
        """The 5th output of netG (reference :187): coloured arg-max offsets of the contextual attention, built on demand from the
        indices the last forward left on the device (nothing in the train step consumes it)."""
        from .inpaint_networks import offsets_to_flow
        P = getattr(self, '_gplan', None)
        return None if P is None else offsets_to_flow(P.attn.argmax, P.B, P.attn.h, P.attn.w, 2)

    # ---------------------------------------------------------------- forward
    def forward(self):
        L = _lib.get()
        B, _, H, W = self.real_A.shape
        cam_t = self._buf('cam_temp', self.CAM)
        L.call('hv_affine', ptr(cam_t), ptr(self.CAM), ctypes.c_longlong(cam_t.numel()), ctypes.c_float(-1.0), ctypes.c_float(1.0), stream())
        P = self.netG.run_forward(self.real_A, self.mask, cam_t, self.slice_ratio, training=self.netG.training)
        self._gplan = P
        self.coarse_seg_sigmoid, self.fake_B_mask_sigmoid = P.coarse_seg, P.fine_seg
        self.x_stage1, self.fake_B_raw = P.x_stage1, P.x_stage2
        d = L.hv_postg_desc()
        outs = {}
        paired = self.isTrain and self.batch_d and 'dcat1' in self._in
        halves = {'fake_B': (1, 0), 'fake_B_mask_raw': (2, 0), 'fake_B_local': (3, 0), 'real_B_local': (3, 1)} if paired else {}
        for n in ('fake_B', 'fake_B_coarse', 'fake_B_local', 'real_B_local', 'fake_B_mask_raw', 'coarse_seg_binary'):
            if n in halves:      # written by the compositing kernel straight into D_k's [fake | real] input buffer
                k, hi = halves[n]
                outs[n] = self._pair_buffer(k, self.real_B.shape)[hi * B:(hi + 1) * B]
            else:
                outs[n] = self._buf(n, self.real_B)
            setattr(self, n, outs[n])
        p1h, p2h = self._buf('pred1_h', shape=(1, B)), self._buf('pred2_h', shape=(1, B))
        self._rows = self._buf('rows', shape=(B, 4), dtype=torch.int32)
        for f, t in (('real_B', self.real_B), ('mask', self.mask), ('x_stage1', P.x_stage1), ('x_stage2', P.x_stage2),
                     ('fine_seg', P.fine_seg), ('coarse_seg', P.coarse_seg), ('pred1', P.pred1), ('pred2', P.pred2),
                     ('height', self.height), ('x1', self.x1), ('x2', self.x2), ('maxheight', self.maxheight),
                     ('fake_B', outs['fake_B']), ('fake_B_coarse', outs['fake_B_coarse']), ('fake_B_local', outs['fake_B_local']),
                     ('real_B_local', outs['real_B_local']), ('fine_bin', outs['fake_B_mask_raw']), ('coarse_bin', outs['coarse_seg_binary']),
                     ('pred1_h', p1h), ('pred2_h', p2h), ('rows', self._rows)):
            setattr(d, f, ptr(t).value)
        d.B, d.H, d.W, d.half_band = B, H, W, self.half_band
        L.call('hv_post_generator', ctypes.byref(d), stream())
        self.pred1_h, self.pred2_h = p1h, p2h
        self.real_edges = ops.sobel(self.real_B_mask, self._buf('real_edges', self.real_B))
        self.fake_edges = ops.sobel(self.fake_B_mask_raw, self._buf('fake_edges', self.real_B))

    # ---------------------------------------------------------------- discriminator updates
    def _loss_slot(self, i):
        return self._loss_buf[i:i + 1].view(())

    def _backward_D(self, k, fake, real):
        """loss_D_k = 0.5 * (BCE(D_k(fake), 0) + BCE(D_k(real), 1)); backward (reference :267-314).  With batch_d the fake and
        the real pass run as ONE 2B-sample launch sequence whose BatchNorm layers keep separate statistics per half and update
        their running statistics half by half -- arithmetically the reference's two consecutive calls, with twice the work per
        kernel launch."""
        net = getattr(self, 'netD_%d' % k)
        mode = self.opt.gan_mode
        L = _lib.get()
        lf, lr = self._loss_slot(2 * k), self._loss_slot(2 * k + 1)
        if self.batch_d:
            B = fake.shape[0]
            x2 = self._in.get('dcat%d' % k)
            if not (x2 is not None and x2.shape[0] == 2 * B and fake.data_ptr() == x2.data_ptr() and real.data_ptr() == x2[B:].data_ptr()):
                # (callers with tensors of their own: gather the two halves)
                x2 = self._buf('dcat%d' % k, shape=(2 * B,) + tuple(fake.shape[1:]))
                n = ctypes.c_longlong(fake.numel())
                L.call('hv_affine', ptr(x2[:B]), ptr(fake), n, ctypes.c_float(1.0), ctypes.c_float(0.0), stream())
                L.call('hv_affine', ptr(x2[B:]), ptr(real), n, ctypes.c_float(1.0), ctypes.c_float(0.0), stream())
            P = net.run_forward(x2, training=True, prep='if_stale', groups=2)      # (step t + 1 finds the tables step t's generator part laid out after D_k's Adam step)
            net.loss_backward_halves(P, mode, lf, lr, 0.5 * self.grad_scale, dz=self._buf('dzz%d' % k, P.logits))
        else:
            P = net.run_forward(fake, training=True, prep='if_stale')
            dz = self._buf('dz%d' % k, P.logits)
            ops.gan_loss(P.logits, False, mode, loss=lf, dz=dz, grad_weight=0.5 * self.grad_scale)
            net.run_backward(P, dz, need_dx=False, param_grads=True, accumulate=False)
            P = net.run_forward(real, training=True, prep=False)
            ops.gan_loss(P.logits, True, mode, loss=lr, dz=dz, grad_weight=0.5 * self.grad_scale)
            net.run_backward(P, dz, need_dx=False, param_grads=True, accumulate=True)
        net.finish()
        setattr(self, 'loss_D_fake_%d' % k, lf)
        setattr(self, 'loss_D_real_%d' % k, lr)

    def _d_real_first(self, k, real):
        """Real-image half of loss_D_k (reference :285-296), run FIRST: it needs nothing from the generator, so its stream overlaps
        the generator forward.  Gradients are assigned; BatchNorm running statistics use the swapped-order momentum."""
        net = getattr(self, 'netD_%d' % k)
        lr = self._loss_slot(2 * k + 1)
        P = net.run_forward(real, training=True, prep='if_stale', stat_order='swapped_first')
        net.loss_backward(P, True, self.opt.gan_mode, lr, 0.5 * self.grad_scale, need_dx=False, param_grads=True, accumulate=False,
                          dz=self._buf('dz%d' % k, P.logits))
        setattr(self, 'loss_D_real_%d' % k, lr)

    def _d_fake_second(self, k, fake):
        """Fake half of loss_D_k, accumulated onto the real half's gradients (a + b == b + a in IEEE arithmetic: same bits as the
        reference's fake-then-real order)."""
        net = getattr(self, 'netD_%d' % k)
        lf = self._loss_slot(2 * k)
        P = net.run_forward(fake, training=True, prep=False, stat_order='swapped_second')
        net.loss_backward(P, False, self.opt.gan_mode, lf, 0.5 * self.grad_scale, need_dx=False, param_grads=True, accumulate=True,
                          dz=self._buf('dz%d' % k, P.logits))
        net.finish()
        setattr(self, 'loss_D_fake_%d' % k, lf)

    def _real_local_early(self):
        """real_B_local = mask * real_B * centre band (reference :254-258,:263) without waiting for the generator."""
        B, _, H, W = self.real_B.shape
        mc = self._bufs.get(('mc', (B, 1, H, W)))
        if mc is None:
            mc = torch.zeros(B, 1, H, W, dtype=torch.float32, device=self.device)
            mc[:, :, :, W // 2 - self.half_band:W // 2 + self.half_band] = 1
            self._bufs[('mc', (B, 1, H, W))] = mc
        out = self._buf('real_B_local_early', self.real_B)
        n = ctypes.c_longlong(out.numel())
        L = _lib.get()
        L.call('hv_affine', ptr(out), ptr(self.mask), n, ctypes.c_float(1.0), ctypes.c_float(0.0), stream())
        L.call('hv_mul3', ptr(out), ptr(self.real_B), ptr(mc), n, stream())      # (mask * real_B) * band, the reference's order
        return out

    def backward_D_1(self):
        self._backward_D(1, self.fake_B, self.real_B)

    def backward_D_2(self):
        self._backward_D(2, self.fake_B_mask_raw, self.real_B_mask)

    def backward_D_3(self):
        self._backward_D(3, self.fake_B_local, self.real_B_local)

    # ---------------------------------------------------------------- generator update
    def _g_step_D(self, k):
        """D_k(fake_k) with the freshly updated D_k, its share of loss_G_GAN and (k != 2) the gradient wrt fake_k."""
        net = getattr(self, 'netD_%d' % k)
        fake = {1: self.fake_B, 2: self.fake_B_mask_raw, 3: self.fake_B_local}[k]
        P = net.run_forward(fake, training=True, prep=True)
        dz = self._buf('dz%d' % k, P.logits)
        if k != 2:   # D_2 sees a thresholded mask: no gradient path to G (reference :201,:324)
            self._dxs[k] = net.loss_backward(P, True, self.opt.gan_mode, self._loss_slot(15 + k), self.grad_scale / 6.0, need_dx=True, param_grads=False,
                                             loss_weight=1.0 / 6.0, dz=dz)
        else:
            # (only the loss value is wanted; in the fp16 mode the single-launch head serves it -- its gradient goes to the plan's carrier, which nothing reads)
            g = P.g_logits
            if not (networks.LOSS_HEAD and g.f16 and g.t.shape[-1] == 4 and g.coff == 0 and
                    ops.gan_loss_pair(P.logits, True, self._loss_slot(15 + k), ops.Act(g.t, 4, 0), mode=self.opt.gan_mode, loss_weight=1.0 / 6.0,
                                      grad_weight=self.grad_scale / 6.0)):
                ops.gan_loss(P.logits, True, self.opt.gan_mode, loss=self._loss_slot(15 + k), loss_weight=1.0 / 6.0, dz=dz, grad_weight=self.grad_scale / 6.0)

    def backward_G(self, d_done=False):
        L = _lib.get()
        B, _, H, W = self.real_B.shape
        lg = self._loss_slot(0)
        if not d_done:
            self._dxs = {}
            for k in (1, 2, 3):
                self._g_step_D(k)
        dxs = self._dxs
        g = L.hv_gloss_desc()
        seeds = {n: self._buf(n, self.real_B) for n in ('d_fake_B', 'd_fake_B_coarse', 'd_fine_seg', 'd_coarse_seg')}
        dp1, dp2 = self._buf('d_pred1', shape=(B, 1)), self._buf('d_pred2', shape=(B, 1))
        losses = self._loss_buf[8:14]
        for f, t in (('fake_B', self.fake_B), ('fake_B_coarse', self.fake_B_coarse), ('real_B', self.real_B), ('mask', self.mask),
                     ('fine_seg', self.fake_B_mask_sigmoid), ('coarse_seg', self.coarse_seg_sigmoid), ('real_B_mask', self.real_B_mask),
                     ('normal_vert', self.normal_vert), ('fake_edges', self.fake_edges), ('real_edges', self.real_edges),
                     ('pred1_h', self.pred1_h), ('pred2_h', self.pred2_h), ('height', self.height), ('maxheight', self.maxheight),
                     ('losses', losses), ('d_fake_B', seeds['d_fake_B']), ('d_fake_B_coarse', seeds['d_fake_B_coarse']),
                     ('d_fine_seg', seeds['d_fine_seg']), ('d_coarse_seg', seeds['d_coarse_seg']), ('d_pred1', dp1), ('d_pred2', dp2)):
            setattr(g, f, ptr(t).value)
        g.lambda_L1 = float(self.opt.lambda_L1)
        g.grad_scale = self.grad_scale
        # loss_G_GAN = the three discriminators' terms, loss_G, and the discriminator's gradient wrt fake_B added to its seed: folded into the loss
        # kernels (they were six 1-element / 4-MB launches in a row at the head of the generator backward)
        self.loss_G = self._loss_slot(14)
        g.gan_terms, g.n_gan_terms = ptr(self._loss_buf[16:19]).value, 3
        g.loss_G_GAN, g.loss_G = ptr(lg).value, ptr(self.loss_G).value
        g.add_d_fake_B = ptr(dxs[1]).value
        g.B, g.H, g.W = B, H, W
        need = L.size('hv_generator_losses_workspace_bytes', B)
        ws, _ = ops._ws(need, self.device, slot=1)
        g.workspace, g.workspace_bytes = ptr(ws).value, ws.numel()
        L.call('hv_generator_losses', ctypes.byref(g), stream())
        self.loss_G_GAN = lg
        self.loss_G_maskL1, self.loss_G_Dice, self.loss_coarse_Dice = losses[0], losses[1], losses[2]
        self.loss_edge, self.loss_h = losses[3], losses[4]
        # gradient wrt the composited images -> wrt the raw generator outputs (rows [xu, xb) only)
        d_x2, d_x1 = self._buf('d_x_stage2', self.real_B), self._buf('d_x_stage1', self.real_B)
        L.call('hv_shrm_backward', ptr(seeds['d_fake_B']), ptr(dxs[3]), ptr(self.mask), ptr(self._rows), 0, ptr(d_x2), B, H, W,
               self.half_band, 0, stream())
        L.call('hv_shrm_backward', ptr(seeds['d_fake_B_coarse']), None, None, ptr(self._rows), 1, ptr(d_x1), B, H, W, self.half_band, 0, stream())
        self.netG.run_backward(self._gplan, seeds['d_coarse_seg'], seeds['d_fine_seg'], d_x1, d_x2, dp1, dp2)

    # ---------------------------------------------------------------- the step, in three device-only phases
    def _phase_a(self):
        """forward; D_1, D_2, D_3 forward/backward (reference :356-370 up to the optimiser steps).  The three discriminator
        updates are independent of each other, so each runs on its own HIP stream (kernels of different discriminators
        overlap on the 256 CUs)."""
        main = torch.cuda.current_stream(self.device)
        if getattr(self, '_d_streams', None) is None:
            self._d_streams = [engine.named_stream('discriminator-%d' % k, self.device) for k in (1, 2, 3)]
            engine.NO_FORK_STREAMS.update(st.cuda_stream for st in self._d_streams)
        self._dxs = {}
        if self._inline_exchange:
            self.grad_sync.chain_reset()      # the step's collectives form one chain D_1 -> D_2 -> D_3 -> G (ddp.GradSync.reduce_branch)
            if __import__('os').environ.get('HV_DP_BRANCH', 'chain') == 'stream':
                # the exchange branch leaves the MAIN stream here, before the discriminator streams do (a first-level fork of the capture)
                self.grad_sync.exchange_stream(self.device).wait_stream(main)
        split = self.real_first and not self.batch_d
        def on(k):
            side = self._d_streams[k - 1] if (self.concurrent_d and not engine.SERIAL) else main
            if side is not main:
                side.wait_stream(main)
            return side
        if split:      # the discriminators' real-image passes do not depend on the generator: they start now, on their streams
            reals = {1: self.real_B, 2: self.real_B_mask, 3: self._real_local_early()}
            for k in (1, 2, 3):
                with torch.cuda.stream(on(k)):
                    self.set_requires_grad(getattr(self, 'netD_%d' % k), True)
                    getattr(self, 'optimizer_D_%d' % k).zero_grad()
                    self._d_real_first(k, reals[k])
        # (Measured and not kept: the batched passes' weight tables laid out on the discriminator streams BEFORE the generator forward -- the early
        # fork of the three streams in the step graph costs more than the 25 us it hides: 7.89 -> 8.05-8.38 ms over four same-box pairs.)
        self.forward()
        fakes = {1: self.fake_B, 2: self.fake_B_mask_raw, 3: self.fake_B_local}
        for k, bw in ((1, self.backward_D_1), (2, self.backward_D_2), (3, self.backward_D_3)):
            with torch.cuda.stream(on(k)):
                if split:
                    self._d_fake_second(k, fakes[k])
                else:
                    self.set_requires_grad(getattr(self, 'netD_%d' % k), True)
                    getattr(self, 'optimizer_D_%d' % k).zero_grad()
                    bw()
                if self._inline_exchange and self.dp_schedule == 'overlapped':       # D_k's mean over the ranks, beside the other discriminators' passes
                    self.grad_sync.reduce_branch(getattr(self, 'netD_%d' % k).paramset().flat_grad)
        if not self._through_ab:
            self._join_d(main)
        if self._inline_exchange and self.dp_schedule != 'overlapped':      # 'captured': the three discriminators' gradients as ONE collective where their streams join
            arena = getattr(self, '_d_grad_arena', None)
            for f in ([arena] if arena is not None else [getattr(self, 'netD_%d' % k).paramset().flat_grad for k in (1, 2, 3)]):
                self.grad_sync.reduce_branch(f)

    def _phase_b(self):
        """D_k optimiser step, D_k forward on the fakes with the updated weights (its own stream), generator losses and
        backward (reference :370-382 up to optimizer_G.step)."""
        main = torch.cuda.current_stream(self.device)
        for k in (1, 2, 3):
            side = self._d_streams[k - 1] if (self.concurrent_d and not engine.SERIAL) else main
            if side is not main and not self._through_ab:
                side.wait_stream(main)
            with torch.cuda.stream(side):
                self._opt_step(getattr(self, 'optimizer_D_%d' % k), getattr(self, 'netD_%d' % k))
                self._g_step_D(k)
        self._join_d(main)
        self.set_requires_grad([self.netD_1, self.netD_2, self.netD_3], False)
        self.optimizer_G.zero_grad()
        self.backward_G(d_done=True)
        if self._inline_exchange:
            self.grad_sync.reduce_branch(self.netG.paramset().flat_grad)

    def _phase_c(self):
        self._opt_step(self.optimizer_G, self.netG)

    def _opt_step(self, optimizer, net):
        """Adam step; in the fp16 storage mode behind the device-side overflow guard: the scaled gradients of a step may overflow an fp16 gradient
        buffer (inf / nan), which then reach every parameter gradient of the network -- such a step is skipped (weights, moments, step count
        unchanged; under data parallelism the check runs on the reduced gradient, so every rank takes the same decision) and counted
        (overflow_steps()).  The scale itself is static (HV_GRAD_SCALE, a power of two; head room in DESIGN.md section 3)."""
        optimizer.step(sync_lr=False, guard_flat=net.paramset().flat_grad if self.grad_scale != 1.0 else None, grad_mul=1.0 / self.grad_scale)
        net.paramset().weights_changed()

    def overflow_steps(self):
        """{network: optimiser steps skipped by the overflow guard so far} (a host read)."""
        self.sync_tail()
        return {n: getattr(self, 'optimizer_' + n).skipped_steps() for n in ('G', 'D_1', 'D_2', 'D_3')}

    OVERFLOW_WARN_RUN = 3      # consecutive loss reads that each saw new skipped steps before the scale is called too large

    def get_current_losses(self):
        """The reference's loss dict (base_model.py:136-142).  The losses are read on the host here anyway, so the overflow guard's counters are read
        with them: a step skipped by the guard is reported on the spot, and skips seen at OVERFLOW_WARN_RUN reads in a row say that HV_GRAD_SCALE is
        too large for this data (a persistent overflow would otherwise freeze a network's weights behind normal-looking losses)."""
        out = BaseModel.get_current_losses(self)
        if self.isTrain and self.grad_scale != 1.0:
            now = self.overflow_steps()
            last = getattr(self, '_overflow_seen', None) or dict.fromkeys(now, 0)
            new = {n: now[n] - last[n] for n in now if now[n] > last[n]}
            self._overflow_seen = now
            self._overflow_run = getattr(self, '_overflow_run', 0) + 1 if new else 0
            if new:
                import warnings
                msg = 'fp16 overflow guard skipped optimiser steps since the last loss read: %s (HV_GRAD_SCALE=%g)' % (new, self.grad_scale)
                if self._overflow_run >= self.OVERFLOW_WARN_RUN:
                    msg += ' -- %d reads in a row: the gradient scale is too large for this data, restart with HV_GRAD_SCALE=%g' % (self._overflow_run, self.grad_scale / 4)
                warnings.warn(msg)
        return out

    def _join_d(self, main):
        if self.concurrent_d and not engine.SERIAL:
            for side in self._d_streams:
                main.wait_stream(side)

    GRAPH_WARMUP = 2     # eager steps before capture (lazy allocations, stream creation, weight tables)

    def _graph_failed(self, e):
        """hipGraph capture refused by the runtime: keep launching eagerly and say so once -- unless the caller asked for a
        hard failure (bench.py: a number labelled 'hipGraph replay' must never come from eager launches)."""
        if getattr(self, 'strict_graph', False):
            raise RuntimeError('hipGraph capture of the train step failed: %s' % str(e).splitlines()[0]) from e
        import warnings
        warnings.warn('hipGraph capture of the train step failed (%s); continuing with eager launches' % str(e).splitlines()[0])
        self.use_graph, self._graphs = False, None
        torch.cuda.synchronize(self.device)

    def optimize_parameters(self):
        """forward; D_1, D_2, D_3 updates; G update (reference :356-382).

        The step is device-only (no host reads, learning rate and Adam step count live on the device), so after
        GRAPH_WARMUP eager steps it is captured once as a hipGraph and replayed: ~490 kernel launches per step become
        one graph launch, which removes the host launch latency that otherwise leaves the GPU idle between the short
        kernels of the backward passes.  In a multi-GPU job (one process per GPU) the four networks' flat gradients are
        averaged over the ranks inside the same step -- see `dp_schedule` in __init__; HV_GRAPH=0 or an active kernel
        timer keeps the eager path."""
        # after the first eager step for this batch shape every convolution of the four networks has been dispatched once: from then on the weight
        # layout passes write only the tables those kernels read (engine.lean_tables)
        if self.isTrain and self.dp_schedule == 'auto':
            self._resolve_dp_schedule()
        with engine.lean_tables(self._eager_steps >= 1):
            return self._optimize_parameters()

    # ---------------------------------------------------------------- data-parallel schedule: preflight
    def _resolve_dp_schedule(self):
        """'auto' -> a schedule, once, at the first step: single process -> nothing to choose; otherwise dp_preflight() over the schedules the transport can
        run (gloo cannot be captured: 'graphs' only -- the preflight then still proves that every rank ends with the same weights and records the time)."""
        if not self.grad_sync.active():
            self.dp_schedule = 'graphs'
            return
        import os
        if os.environ.get('HV_DP_PREFLIGHT', '1') == '0':
            self.dp_schedule = 'captured' if self.grad_sync.capturable() else 'graphs'
            return
        self.dp_preflight(schedules=('graphs', 'captured', 'overlapped') if self.grad_sync.capturable() else ('graphs',))

    def _dp_state(self):
        """Every tensor a train step changes besides the activations: the four networks' parameters and buffers (BatchNorm running statistics,
        spectral-norm vectors), the optimisers' moments / step counts / overflow counters, the loss slots."""
        ts = []
        for n in ('G', 'D_1', 'D_2', 'D_3'):
            net = getattr(self, 'net' + n)
            ts += [p.data for p in net.parameters()] + list(net.buffers())
        for o in self.optimizers:
            o._ensure_state()
            ts += [o._m, o._v, o._step]
        ts.append(self._loss_buf)
        return ts

    def _dp_weight_checksum(self):
        """Order-independent, exact checksum of all four networks' weights: the int64 sum of their fp32 bit patterns."""
        tot = torch.zeros((), dtype=torch.int64, device=self.device)
        for n in ('G', 'D_1', 'D_2', 'D_3'):
            for p in getattr(self, 'net' + n).parameters():
                tot += p.data.view(torch.int32).to(torch.int64).sum()
        return tot

    def dp_preflight(self, timed_steps=5, schedules=('graphs', 'captured', 'overlapped')):
        """Pick the data-parallel schedule on THIS job, on the batch set_input() just delivered, before the first training step.

        Both schedules ('captured': the exchange branch inside the step's one hipGraph; 'graphs': the step cut at the exchanges) are run from the
        same weights: GRAPH_WARMUP eager steps, the capture, two replays, then `timed_steps` replays between barriers.  After each the ranks compare
        (a) that the schedule ran on every rank, (b) an exact checksum of all weights (MIN == MAX over the ranks: the collectives delivered the same
        mean to every rank, in the same order), (c) the slowest rank's time.  The faster schedule that passed is kept -- with its captured graphs --
        and every tensor the steps touched (weights, running statistics, Adam state) is put back: preflight steps are not training steps.
        A schedule that raises, diverges across the ranks or is refused by the runtime is recorded and dropped; if none is left the job stops with
        the recorded text.  A rank that never comes back from a collective cannot be recovered in-process: a timer (HV_DP_PREFLIGHT_TIMEOUT_S,
        default 300 s) then ends THIS process with the text on stderr and exit code 3 instead of hanging the launcher (never a re-exec: the
        process has touched the GPU; a retry is a fresh job)."""
        import os
        import sys
        import threading
        import time
        import torch.distributed as dist
        world = dist.get_world_size()
        rec = {'world_size': world, 'timed_steps': timed_steps, 'schedules': {}, 'chosen': None}
        self.dp_preflight_record = rec
        limit = float(os.environ.get('HV_DP_PREFLIGHT_TIMEOUT_S', '300'))
        where = {'at': 'start'}

        def expired():
            sys.stderr.write('healthivert-gan_amd: data-parallel preflight did not finish within %.0f s (rank %d, in %s): a rank is stuck in a collective; '
                             'giving up (exit 3).  Record so far: %r\n' % (limit, dist.get_rank(), where['at'], rec))
            sys.stderr.flush()
            os._exit(3)
        timer = threading.Timer(limit, expired)
        timer.daemon = True
        timer.start()
        self._home_d_grads()
        for net in (self.netG, self.netD_1, self.netD_2, self.netD_3):      # (gradient storage and tables in place before the snapshot)
            net.paramset()._ensure(self.device)
        state = self._dp_state()
        snap = [t.clone() for t in state]
        kept = {}
        self._in_preflight = True
        try:
            for sched in schedules:      # (the plain one first: its captured graphs are kept whatever the later trials do)
                where['at'] = sched
                r = {'ok': False, 'error': None, 'ms_per_step': None, 'weights_identical_across_ranks': None}
                rec['schedules'][sched] = r
                self.dp_schedule, self._graphs, self._eager_steps = sched, None, 0
                self.dp_capture_error = None
                ok, dt = 1.0, float('inf')
                try:
                    with engine.lean_tables(False):
                        self._optimize_parameters()
                    for _ in range(self.GRAPH_WARMUP + 2):
                        with engine.lean_tables(True):
                            self._optimize_parameters()
                    if self.use_graph and self._graphs is None:
                        raise RuntimeError('the step was not captured')
                    torch.cuda.synchronize(self.device)
                    dist.barrier()
                    torch.cuda.synchronize(self.device)
                    t0 = time.perf_counter()
                    for _ in range(timed_steps):
                        with engine.lean_tables(True):
                            self._optimize_parameters()
                    torch.cuda.synchronize(self.device)
                    dt = (time.perf_counter() - t0) / timed_steps
                except Exception as e:      # noqa: BLE001 -- recorded; the other schedule may still serve
                    ok, r['error'] = 0.0, '%s: %s' % (type(e).__name__, (str(e).splitlines() or ['?'])[0])
                    torch.cuda.synchronize(self.device)
                # ---- what the other ranks saw (these small collectives run in every case, so that a failure on one rank cannot strand the others)
                agg = torch.tensor([ok, -dt if ok else 0.0], dtype=torch.float64, device=self.device)
                dist.all_reduce(agg, op=dist.ReduceOp.MIN)
                ck = self._dp_weight_checksum()
                ck2 = torch.stack([ck, -ck])
                dist.all_reduce(ck2, op=dist.ReduceOp.MIN)
                torch.cuda.synchronize(self.device)
                all_ok = float(agg[0].item()) == 1.0
                same = int(ck2[0].item()) == -int(ck2[1].item())
                r['weights_identical_across_ranks'] = same
                if all_ok:
                    r['ms_per_step'] = round(-float(agg[1].item()) * 1e3, 3)
                elif r['error'] is None:
                    r['error'] = 'failed on another rank'
                r['ok'] = bool(all_ok and same)
                if r['ok']:
                    kept[sched] = (self._graphs, self._eager_steps)
                for t, c in zip(state, snap):      # the same starting point for the next schedule, and for training
                    t.copy_(c)
                for k in (1, 2, 3):      # the discriminators' prepared tables belong to the weights just overwritten, and the captured steps (rightly) no longer lay
                    ps = getattr(self, 'netD_%d' % k).paramset()      # them out before the discriminator update: once, here, from the restored weights
                    ps.weights_changed()
                    ps.prep(self.device, False)
                torch.cuda.synchronize(self.device)
        finally:
            self._in_preflight = False
            timer.cancel()
        good = [k for k in schedules if rec['schedules'][k]['ok']]
        if not good:
            raise RuntimeError('data-parallel preflight: no schedule ran correctly on %d rank(s): %r' % (world, rec['schedules']))
        best = min(good, key=lambda k: rec['schedules'][k]['ms_per_step'])
        for pref in ('captured', 'overlapped'):      # (within a percent of the fastest: no graph cut / the collectives beside compute)
            if pref in good and rec['schedules'][pref]['ms_per_step'] <= 1.01 * rec['schedules'][best]['ms_per_step']:
                best = pref
        rec['chosen'] = best
        self.dp_schedule = best
        self._graphs, self._eager_steps = kept[best]
        self.dp_capture_error = next((rec['schedules'][k]['error'] for k in ('captured', 'overlapped') if k in rec['schedules'] and rec['schedules'][k]['error']), None)
        if dist.get_rank() == 0:
            print('data-parallel preflight (%d rank(s)): %s -> %s' % (world, {k: (v['ms_per_step'], v['error']) for k, v in rec['schedules'].items()}, best), flush=True)

    def _optimize_parameters(self):
        for o in self.optimizers:
            o.sync_lr()
        graphable = self.use_graph and ops.timer() is None
        dp = self.grad_sync.active()
        inline = dp and self.dp_schedule in ('captured', 'overlapped') and self.grad_sync.capturable()
        self._inline_exchange = inline
        cut = dp and not inline            # the means sit BETWEEN the step's graphs (exchange stream, issued eagerly)
        if dp and self._eager_steps == 0 and self._graphs is None:
            self._home_d_grads()      # (both schedules: a preflight runs them over the same gradient storage, and captured graphs keep its addresses)
        if graphable and self._graphs is None and self._eager_steps >= self.GRAPH_WARMUP:
            try:
                self._capture(cut)
            except RuntimeError as e:
                if inline and self._in_preflight:
                    raise
                if inline:
                    # a runtime that refuses to capture the collectives: keep the step, cut it at the exchanges instead (they are then issued eagerly)
                    import warnings
                    warnings.warn('hipGraph capture of the step with its RCCL collectives failed (%s); falling back to HV_DP_SCHEDULE=graphs' % str(e).splitlines()[0])
                    torch.cuda.synchronize(self.device)
                    self.dp_schedule, self._graphs, self._eager_steps = 'graphs', None, 0
                    self.dp_capture_error = str(e).splitlines()[0]
                    return self.optimize_parameters()
                self._graph_failed(e)
                graphable = False
        replay = graphable and self._graphs is not None
        if replay and len(self._graphs) == 1:      # the whole step as one graph
            self._graphs[0].replay()
            return
        phases = self._graphs if replay else (self._phase_a, self._phase_b, self._phase_c)
        run = (lambda ph: ph.replay()) if replay else (lambda ph: ph())
        run(phases[0])
        if cut:
            self._exchange([self.netD_1, self.netD_2, self.netD_3])
        run(phases[1])
        if cut:
            self._exchange([self.netG])
        run(phases[2])
        if not replay:
            self._eager_steps += 1

    def _home_d_grads(self):
        """Cut schedule: the three discriminators' flat gradient buffers as three consecutive slices of ONE buffer (ParamSet.grad_home, taken up
        when a set lays out its tables), so that their means are one collective."""
        if getattr(self, '_d_grad_arena', None) is not None:
            return
        sets = [getattr(self, 'netD_%d' % k).paramset() for k in (1, 2, 3)]
        sizes = [sum(p.numel() for p in ps.trainable()) for ps in sets]
        rup = lambda n: (n + 63) // 64 * 64          # every slice starts on a 256-byte boundary (the guarded Adam's vector loads); the gaps stay zero
        self._d_grad_arena = torch.zeros(sum(rup(n) for n in sizes), dtype=torch.float32, device=self.device)
        off = 0
        for ps, n in zip(sets, sizes):
            ps.grad_home = self._d_grad_arena[off:off + n]
            ps._key = None          # lay the tables out again over the new gradient storage at the next prep
            for t in list(ps.t_prep.values()) + list(ps.t_bwd.values()):
                t.key = None        # (their rows hold pointers into the gradient storage)
            off += rup(n)

    def _exchange(self, nets):
        """Cut schedule: mean over the ranks of the networks' flat gradients (one all-reduce each, issued back to back on the exchange stream -- ONE
        for all of them when their buffers lie back to back, see _home_d_grads); the current stream continues when all of them are done.  Nothing
        blocks the host."""
        main = torch.cuda.current_stream(self.device)
        flats = [n.paramset().flat_grad for n in nets]
        arena = getattr(self, '_d_grad_arena', None)
        if len(flats) == 3 and arena is not None:
            lo, hi = arena.data_ptr(), arena.data_ptr() + 4 * arena.numel()
            if all(lo <= f.data_ptr() and f.data_ptr() + 4 * f.numel() <= hi for f in flats) and sum((f.numel() + 63) // 64 * 64 for f in flats) == arena.numel():
                flats = [arena]          # the three discriminators' buffers (and the zero gaps between them) as one collective
        events = [self.grad_sync.reduce(f, after=main) for f in flats]
        for ev in events:
            if ev is not None:
                main.wait_event(ev)

    def sync_tail(self):
        """Nothing of a step runs on after optimize_parameters() returns to the current stream (the twelve-phase schedule of rounds 1-3, whose
        generator Adam step trailed on the exchange stream, is gone); kept for callers."""
        return None

    def _capture(self, cut=False):
        if getattr(self, '_capture_stream', None) is None:
            self._capture_stream = engine.named_stream('capture', self.device)
        torch.cuda.synchronize(self.device)
        if self.grad_sync.active() and self.grad_sync.capturable():
            # torch's ProcessGroupNCCL retires finished collectives from its watchdog thread (a poll every 100 ms) by querying their events.  On this stack a
            # query that lands while ANY stream of the process is capturing can fail with hipErrorCapturedEvent ("operation not permitted on an event last
            # recorded in a capturing stream" -- torch draws its collective stream from the same pool of 32 streams per device as the step's streams), and
            # the watchdog then aborts the process: seen once in ~20 runs of the one-rank RCCL test, a few ms after the warm-up steps' collectives (weight
            # broadcast, the communicator's id exchange, the cut schedule's means).  The device is idle here: give the watchdog time for three polls so
            # that nothing is left for it to query during the capture.  Once per batch shape.
            import os
            import time
            time.sleep(float(os.environ.get('HV_DP_CAPTURE_SETTLE_MS', '350')) * 1e-3)
        graphs, pool = [], None
        one = not cut      # (the cut data-parallel schedule issues its gradient means between the graphs)

        def whole():
            # one graph: D_k goes from its backward straight on to its Adam step and its pass for the generator on its own stream -- no join of the three
            # discriminator streams between the phases (that join only exists for the cut schedule's exchange): a discriminator that is done early
            # (D_3 reads the 128 x 128 crop) does not wait for the others
            self._through_ab = self.concurrent_d and not engine.SERIAL and not (self._inline_exchange and self.dp_schedule == 'captured')
            try:
                self._phase_a(); self._phase_b(); self._phase_c()
            finally:
                self._through_ab = False
        for phase in ((whole,) if one else (self._phase_a, self._phase_b, self._phase_c)):
            g = torch.cuda.CUDAGraph()
            # thread_local: a process-group watchdog thread polling its events must not invalidate the capture
            with torch.cuda.graph(g, pool=pool, stream=self._capture_stream, capture_error_mode='thread_local'):
                phase()
            pool = g.pool()
            graphs.append(g)
        self._graphs = tuple(graphs)
