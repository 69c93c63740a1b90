"""Generator / ContextualAttention / discriminator on the HIP path (fp32 mode) against the golden vectors the
reference produced, at the north-star tolerance |d| <= 1e-3 (activations) -- observed errors are ~1e-5."""
import os

import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _err(a, b):
    return (a.detach().cpu().double() - b.double()).abs().max().item()


def _flow_close(got, ref, name):
    """offset_flow: uint8 colour codes / 255.  The arg-max of near-tied scores may pick a different patch on the device than the CPU
    reference did (a different colour for that pixel's 8x8 block), and floor(255*col) can fall either side of an integer: allow a
    handful of pixels; everything else must agree to half a colour step."""
    d = (got.detach().cpu() - ref).abs()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    frac = (d > 0.5 / 255).float().mean().item()
    assert frac <= 5e-3, (name, frac, d.max().item())


def _gen(g, ngf=4):
    import hvgan
    from hvgan.models.inpaint_networks import Generator
    net = Generator({'input_dim': 1, 'ngf': ngf}, True)
    net.load_state_dict(g['sd'])
    net.cuda()
    net.precision = 'fp32'
    return net


def test_g1_generator_forward_buffers_and_grads():
    g = load_golden('g1_generator_mini')
    net = _gen(g)
    net.train()
    dev = torch.device('cuda:0')
    x, mask, cam, ratio = (g[k].to(dev) for k in ('x', 'mask', 'cam', 'ratio'))
    P = net.run_forward(x, mask, cam, ratio, training=True)
    torch.cuda.synchronize()
    outs = dict(coarse_seg=P.coarse_seg, fine_seg=P.fine_seg, x_stage1=P.x_stage1, x_stage2=P.x_stage2, pred1_h=P.pred1, pred2_h=P.pred2)
    for n, t in outs.items():
        assert _err(t, g['train'][n]) <= TOL, (n, _err(t, g['train'][n]))
    sd = net.state_dict()
    for k, v in g['bufs_after'].items():
        assert _err(sd[k], v) <= 1e-4, k
    seeds = [g['coef'][str(i)].to(dev) for i in range(6)]
    net.run_backward(P, seeds[0], seeds[1], seeds[2], seeds[3], seeds[4], seeds[5])
    torch.cuda.synchronize()
    worst = 0.0
    for k, p in net.named_parameters():
        ref = g['grads'][k]
        e = _err(p.grad, ref) / max(1.0, ref.abs().max().item())
        worst = max(worst, e)
        assert e <= TOL, (k, e)
    net.eval()
    P = net.run_forward(x, mask, cam, ratio, training=False)
    torch.cuda.synchronize()
    outs = dict(coarse_seg=P.coarse_seg, fine_seg=P.fine_seg, x_stage1=P.x_stage1, x_stage2=P.x_stage2, pred1_h=P.pred1, pred2_h=P.pred2)
    for n, t in outs.items():
        assert _err(t, g['eval'][n]) <= TOL, (n, _err(t, g['eval'][n]))
    # offset_flow slot (reference :368,:389-410): train- and eval-mode values against the reference's own tensors
    from hvgan.models.inpaint_networks import offsets_to_flow
    _flow_close(offsets_to_flow(P.attn.argmax, P.B, P.attn.h, P.attn.w, 2), g['flow_eval'], 'flow_eval')
    net.load_state_dict(g['sd'])
    net.train()
    P = net.run_forward(x, mask, cam, ratio, training=True)
    _flow_close(offsets_to_flow(P.attn.argmax, P.B, P.attn.h, P.attn.w, 2), g['flow_train'], 'flow_train')


def test_g1_generator_module_call_and_autograd_bridge():
    g = load_golden('g1_generator_mini')
    net = _gen(g)
    net.train()
    dev = torch.device('cuda:0')
    x, mask, cam, ratio = (g[k].to(dev) for k in ('x', 'mask', 'cam', 'ratio'))
    o = net(x, mask, cam, ratio)
    assert len(o) == 7 and o[4].shape == (2, 3, 64, 64)
    _flow_close(o[4], g['flow_train'], 'flow_train (module call)')
    assert float(o[4].abs().max()) > 0
    outs = [o[0], o[1], o[2], o[3], o[5], o[6]]
    loss = sum((a * g['coef'][str(i)].to(dev)).sum() for i, a in enumerate(outs))
    assert abs(loss.item() - g['loss'].item()) <= 1e-2 * max(1.0, abs(g['loss'].item()))
    loss.backward()
    for k, p in net.named_parameters():
        ref = g['grads'][k]
        assert _err(p.grad, ref) <= TOL * max(1.0, ref.abs().max().item()), k


def test_g2_contextual_attention_batch0_mask_quirk():
    from hvgan.models.inpaint_networks import ContextualAttention
    g = load_golden('g2_attention')
    dev = torch.device('cuda:0')
    ca = ContextualAttention(True, ksize=3, stride=1, rate=2, fuse_k=3, softmax_scale=10, fuse=True)
    f = g['f'].to(dev).requires_grad_(True)
    y, flow = ca(f, f, g['mask'].to(dev))
    assert _err(y, g['y']) <= TOL
    _flow_close(flow, g['flow'], 'g2 flow')
    (y * g['coef'].to(dev)).sum().backward()
    assert _err(f.grad, g['grad_f']) <= TOL * max(1.0, g['grad_f'].abs().max().item())


@pytest.mark.parametrize('fixture,netD,n_layers,norm', [('g3_disc_batch', 'basic', 3, 'batch'), ('g3_disc_instance', 'basic', 3, 'instance'),
                                                        ('g3n_disc_n2_batch', 'n_layers', 2, 'batch'), ('g3n_disc_n4_batch', 'n_layers', 4, 'batch'),
                                                        ('g3n_disc_n4_instance', 'n_layers', 4, 'instance')])
def test_g3_discriminator(fixture, netD, n_layers, norm):
    """PatchGAN forward / loss / every gradient / running statistics after three calls / eval forward against the reference's own outputs: 'basic' (G3) and
    define_D('n_layers', n_layers_D in {2, 4}) (G3n; reference models/networks.py:198-199 -- the depths 'basic' does not build)."""
    from hvgan.models import networks
    g = load_golden(fixture)
    dev = torch.device('cuda:0')
    net = networks.define_D(1, 8, netD, n_layers, norm, 'normal', 0.02, [])
    net.load_state_dict(g['sd'])
    net.cuda()
    net.precision = 'fp32'
    net.train()
    x0 = g['x']['0'].to(dev)
    P = net.run_forward(x0, training=True)
    assert _err(P.logits, g['y']['0']) <= TOL
    loss = torch.zeros((), device=dev)
    dz = torch.empty_like(P.logits)
    from hvgan import ops
    ops.gan_loss(P.logits, True, 'vanilla', loss=loss, dz=dz)
    assert abs(loss.item() - g['loss'].item()) <= 1e-4
    dx = net.run_backward(P, dz, need_dx=True, param_grads=True)
    net.finish()
    torch.cuda.synchronize()
    for k, p in net.named_parameters():
        ref = g['grads'][k]
        assert _err(p.grad, ref) <= TOL * max(1.0, ref.abs().max().item()), k
    assert _err(dx, g['grad_x']) <= TOL * max(1.0, g['grad_x'].abs().max().item())
    for i in (1, 2):
        P = net.run_forward(g['x'][str(i)].to(dev), training=True)
        assert _err(P.logits, g['y'][str(i)]) <= TOL
    sd = net.state_dict()
    for k, v in g.get('sd_after', {}).items():
        assert _err(sd[k].double(), v) <= 1e-4, k
    net.eval()
    assert _err(net(x0), g['y_eval']) <= TOL


def test_g6_unet_ct_mask_forward_backward():
    from hvgan.models.UnetG_CT_mask import define_G
    g = load_golden('g6_unet_mini')
    dev = torch.device('cuda:0')
    net = define_G(3, 1, 4, 'unet_256', 'batch', False, 'normal', 0.02, [])
    net.load_state_dict(g['sd'])
    net.cuda()
    net.precision = 'fp32'
    net.train()
    x, tgt = g['x'].to(dev), g['tgt'].to(dev)
    ct, mk = net(x)
    assert _err(ct, g['ct']) <= TOL and _err(mk, g['mk']) <= TOL
    loss = (ct - tgt).abs().mean() + (mk * tgt).mean()
    assert abs(loss.item() - g['loss'].item()) <= 1e-3
    loss.backward()
    for k, p in net.named_parameters():
        ref = g['grads'][k]
        assert _err(p.grad, ref) <= TOL * max(1.0, ref.abs().max().item()), (k, _err(p.grad, ref))
    sd = net.state_dict()
    for k, v in g['sd_after'].items():
        assert _err(sd[k].double(), v) <= 1e-4, k
    net.eval()
    with torch.no_grad():
        cte, mke = net(x)
    assert _err(cte, g['ct_eval']) <= TOL and _err(mke, g['mk_eval']) <= TOL


def test_g6b_unet_use_dropout_true_quirk():
    """UnetG_CT_mask with use_dropout=True: the reference hands the boolean to nn.Dropout (p = 1.0), so in train mode the inner blocks output zeros and
    in eval mode the network is the plain one (fixture G6b = the reference's outputs).  The backward of that train mode is all zeros through those
    blocks; the HIP path refuses it loudly rather than computing something else."""
    from hvgan.models.UnetG_CT_mask import define_G
    g = load_golden('g6b_unet_dropout')
    dev = torch.device('cuda:0')
    net = define_G(3, 1, 4, 'unet_256', 'batch', True, 'normal', 0.02, [])
    net.load_state_dict(g['sd'])
    net.cuda()
    net.precision = 'fp32'
    net.train()
    x = g['x'].to(dev)
    with torch.no_grad():
        ct, mk = net(x)
    assert _err(ct, g['ct']) <= TOL and _err(mk, g['mk']) <= TOL
    sd = net.state_dict()
    for k, v in g['sd_after'].items():
        assert _err(sd[k].double(), v) <= 1e-4, k
    net.eval()
    with torch.no_grad():
        cte, mke = net(x)
    assert _err(cte, g['ct_eval']) <= TOL and _err(mke, g['mk_eval']) <= TOL
    net.train()
    ct, mk = net(x)
    with pytest.raises(NotImplementedError):
        (ct.sum() + mk.sum()).backward()


@pytest.mark.parametrize('norm', ['batch', 'instance'])
def test_discriminator_module_api_two_forwards_then_one_backward(norm):
    """The reference's own discriminator update through the nn.Module API (pix2pix_model.py:267-283): pred_fake = D(fake.detach());
    pred_real = D(real); loss_D = 0.5 * (GAN(pred_fake, False) + GAN(pred_real, True)); loss_D.backward().  Both passes' activations must
    survive until the single backward and their gradients must ADD: compared with autograd through the CPU oracle.  A second
    zero_grad + backward round must assign again (not keep accumulating)."""
    import hvgan  # noqa: F401
    from hvgan.models import networks
    from hvgan.optim import FusedAdam
    from oracle import restate as R
    g = load_golden('g3_disc_%s' % norm)
    dev = torch.device('cuda:0')
    net = networks.define_D(1, 8, 'basic', 3, norm, 'normal', 0.02, [])
    net.load_state_dict(g['sd'])
    net.cuda().train()
    net.precision = 'fp32'
    crit = networks.GANLoss('vanilla').to(dev)
    opt = FusedAdam(net.parameters(), lr=2e-4, betas=(0.5, 0.999))
    sd = {k: v.clone() for k, v in g['sd'].items()}
    params = [k for k in sd if k.endswith('.weight') or k.endswith('.bias')]
    for k in params:
        sd[k].requires_grad_(True)
    for rnd in range(2):
        fake, real = g['x'][str(rnd)], g['x'][str(rnd + 1)]
        opt.zero_grad()
        pf = net(fake.to(dev).detach())
        pr = net(real.to(dev))
        loss = (crit(pf, False) + crit(pr, True)) * 0.5
        loss.backward()
        torch.cuda.synchronize()
        for k in params:
            sd[k].grad = None
        rf, u1 = R.disc_forward(sd, fake, norm, True)
        with torch.no_grad():
            for k, v in u1.items():
                sd[k].copy_(v)
        rr, u2 = R.disc_forward(sd, real, norm, True)
        with torch.no_grad():
            for k, v in u2.items():
                sd[k].copy_(v)
        rloss = (R.gan_loss(rf, False, 'vanilla') + R.gan_loss(rr, True, 'vanilla')) * 0.5
        rloss.backward()
        assert abs(loss.item() - rloss.item()) <= 1e-4
        got = dict(net.named_parameters())
        for k in params:
            ref = sd[k].grad
            assert _err(got[k].grad, ref) <= TOL * max(1.0, ref.abs().max().item()), (rnd, k, _err(got[k].grad, ref))
    # a forward whose plan is reused before its backward must fail loudly, never return another pass's gradients
    from hvgan.models.inpaint_networks import Generator
    gg = load_golden('g1_generator_mini')
    gen = _gen(gg)
    gen.train()
    x, mask, cam, ratio = (gg[k].to(dev) for k in ('x', 'mask', 'cam', 'ratio'))
    o1 = gen(x, mask, cam, ratio)
    o2 = gen(x, mask, cam, ratio)
    with pytest.raises(RuntimeError, match='overwrote'):
        o1[0].sum().backward()
    o2[0].sum().backward()


def test_module_api_backward_in_fp16_storage_mode_scales_its_gradients():
    """The nn.Module autograd bridges in the fp16 storage mode: a loss with small seeds (mean over the logits times 1e-2: ~1e-6 per logit, the
    size of this model's real activation gradients) backpropagated through `loss.backward()`.  Without the bridges' power-of-two gradient scale
    these gradients are subnormal or zero in the fp16 gradient buffers; with it the parameter gradients and the input gradient follow the
    oracle's (relative L2, fp16 operands), also when a second backward accumulates into .grad."""
    import hvgan  # noqa: F401
    from hvgan.models import networks
    from hvgan.optim import FusedAdam
    from oracle import restate as R
    g = load_golden('g3_disc_batch')
    dev = torch.device('cuda:0')
    net = networks.define_D(1, 8, 'basic', 3, 'batch', 'normal', 0.02, [])
    net.load_state_dict(g['sd'])
    net.cuda().train()
    net.precision = 'fp16'
    opt = FusedAdam(net.parameters(), lr=2e-4, betas=(0.5, 0.999))
    sd = {k: v.clone() for k, v in g['sd'].items()}
    params = [k for k in sd if k.endswith('.weight') or k.endswith('.bias')]
    for k in params:
        sd[k].requires_grad_(True)
    x0, x1 = g['x']['0'], g['x']['1']
    opt.zero_grad()
    xa = x0.to(dev).requires_grad_(True)
    (net(xa).mean() * 1e-2).backward()
    (net(x1.to(dev)).mean() * 1e-2).backward()          # accumulates into .grad (scaled up, added to, scaled down again)
    torch.cuda.synchronize()
    xr = x0.clone().requires_grad_(True)
    r0, u = R.disc_forward(sd, xr, 'batch', True)
    with torch.no_grad():
        for k, v in u.items():
            sd[k].copy_(v)
    r1, _ = R.disc_forward(sd, x1, 'batch', True)
    (r0.mean() * 1e-2 + r1.mean() * 1e-2).backward()
    got = dict(net.named_parameters())
    for k in params:
        ref, gk = sd[k].grad, got[k].grad.cpu()
        rel = (gk - ref).norm().item() / max(ref.norm().item(), 1e-30)
        assert rel <= 0.05, (k, rel, ref.norm().item(), gk.norm().item())
    rel = (xa.grad.cpu() - xr.grad).norm().item() / xr.grad.norm().item()
    assert rel <= 0.05 and xr.grad.abs().max().item() < 1e-4, (rel, xr.grad.abs().max().item())


@pytest.mark.parametrize('M,N,K,batch,scaled,split', [(1024, 1024, 576, 8, True, 0), (1024, 576, 1024, 3, False, 0), (200, 68, 64, 2, True, 0),
                                                      (128, 128, 32, 16, False, 0), (256, 1024, 128, 8, False, 64),
                                                      # the fp16 x fp16 products of these take the LDS-DMA kernel (>= 128 tiles of 256 x 256, K % 64 == 0):
                                                      (1024, 576, 1024, 16, True, 0), (512, 1024, 128, 16, False, 64), (1024, 1024, 64, 12, False, 0),
                                                      (1000, 520, 192, 16, True, 0)])
def test_batched_nt_gemm_against_torch(M, N, K, batch, scaled, split):
    """hv_bgemm_nt (the fp16 mode's attention contractions): operands rounded to fp16, fp32 accumulation -- against torch on the same rounded
    operands in fp64; ragged tile edges, the column scale, the XCD batch swizzle (batch % 8 == 0) and the plain mapping.  The fp16 x fp16 form of the
    large shapes runs in bgemm_dma_kernel (256 x 256 tiles on an LDS-DMA ring) and must give the bits of the register-staged kernel (same k order)."""
    import ctypes
    from hvgan import lib
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(batch, M, K, generator=g)
    Bm = torch.randn(batch, N, K, generator=g)
    cs = (torch.rand(batch, N, generator=g) + 0.5) if scaled else None
    C = torch.full((batch, M, N), float('nan'), device=dev)
    Ad, Bd, csd = A.to(dev), Bm.to(dev), (cs.to(dev) if scaled else None)
    for a16, b16 in ((0, 0), (0, 1), (1, 1)):          # fp32 operands converted when staged / fp16 tables copied as they are: the same products
        C.fill_(float('nan'))
        Ax, Bx = (Ad.half() if a16 else Ad), (Bd.half() if b16 else Bd)
        lib.get().call('hv_bgemm_nt', lib.ptr(Ax), a16, K, ctypes.c_longlong(M * K), lib.ptr(Bx), b16, K, ctypes.c_longlong(N * K), lib.ptr(C), N,
                       ctypes.c_longlong(M * N), M, N, K, batch, ctypes.c_float(0.25), lib.ptr(csd), ctypes.c_longlong(N if scaled else 0), split,
                       lib.stream())
        torch.cuda.synchronize()
        if (a16, b16) == (0, 0):
            C0 = C.clone()
        else:
            assert torch.equal(C, C0), (a16, b16)
    Bl = Bm
    if split:       # logical row t * split + c is stored as row c * (N / split) + t
        Bl = Bm.view(batch, split, N // split, K).transpose(1, 2).reshape(batch, N, K)
    ref = 0.25 * torch.bmm(A.half().double(), Bl.half().double().transpose(1, 2))
    if scaled:
        ref = ref * cs.double()[:, None, :]
    err = (C.cpu().double() - ref).abs().max().item()
    assert err <= 2e-5 * K ** 0.5 * max(1.0, ref.abs().max().item()), err        # fp32 accumulation of K exact fp16 x fp16 products


def test_fold_of_stride2_patches_is_conv_transpose_tail():
    """hv_ca_fold against F.fold (= the overlap-add that ends F.conv_transpose2d(stride 2, padding 1) with 4x4 filters), assign and accumulate."""
    import ctypes
    import torch.nn.functional as F
    from hvgan import lib
    dev = torch.device('cuda:0')
    B, H, W, C = 2, 16, 24, 8
    h, w = H // 2, W // 2
    g = torch.Generator().manual_seed(9)
    src = torch.randn(B, h * w, 16, C, generator=g)                      # [b][p][tap][c]
    cols = src.permute(0, 3, 2, 1).reshape(B, C * 16, h * w)             # F.fold wants [b][c * 16 + tap][p]
    ref = 0.25 * F.fold(cols, (H, W), kernel_size=4, stride=2, padding=1)          # (B, C, H, W)
    dst = torch.ones(B, H, W, C, device=dev)
    for acc in (0, 1):
        lib.get().call('hv_ca_fold', lib.ptr(src.to(dev)), lib.ptr(dst), 0, B, H, W, C, C, ctypes.c_float(0.25), acc, lib.stream())
        torch.cuda.synchronize()
        want = ref.permute(0, 2, 3, 1) * (1 + acc)
        assert (dst.cpu() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


def test_g2_contextual_attention_fp16_gemm_route(monkeypatch):
    """fp16 mode: the attention block with its contractions as batched GEMMs (engine.CA_GEMM) against the G2 fixture at the fp16 mode's tolerance and
    against the same block with the contractions as per-sample-filter convolutions (both round their operands to fp16; only the summation order differs)."""
    from hvgan import engine
    from hvgan.models.inpaint_networks import ContextualAttention
    monkeypatch.setenv('HV_PRECISION', 'fp16')
    g = load_golden('g2_attention')
    dev = torch.device('cuda:0')
    outs = {}
    for gemm in (True, False):
        monkeypatch.setattr(engine, 'CA_GEMM', gemm)
        ca = ContextualAttention(True, ksize=3, stride=1, rate=2, fuse_k=3, softmax_scale=10, fuse=True)
        f = g['f'].to(dev).requires_grad_(True)
        y, _ = ca(f, f, g['mask'].to(dev))
        (y * g['coef'].to(dev)).sum().backward()
        outs[gemm] = (y.detach().float().cpu(), f.grad.detach().float().cpu())
    ys, gs = max(1.0, g['y'].abs().max().item()), max(1.0, g['grad_f'].abs().max().item())
    for gemm in (True, False):
        assert _err(outs[gemm][0], g['y']) <= 2e-2 * ys, (gemm, _err(outs[gemm][0], g['y']))
        assert _err(outs[gemm][1], g['grad_f']) <= 4e-2 * gs, (gemm, _err(outs[gemm][1], g['grad_f']))
    assert _err(outs[True][0], outs[False][0]) <= 5e-3 * ys
    assert _err(outs[True][1], outs[False][1]) <= 1e-2 * gs


@pytest.mark.parametrize('adjoint', [0, 1])
def test_score_fusion_tiles_match_the_index_formula(adjoint):
    """hv_ca_fuse on the 32 x 32 map (LDS-tiled kernels) against the defining double sum evaluated with numpy in the same order of additions:
    forward  out[p][l] = sum_d sum_e S[itr(tr(p)+d)+e][itr(tr(l)+d)+e],  adjoint  out[p][l] = sum_e sum_d S[itr(tr(p+e)+d)][itr(tr(l+e)+d)]
    (terms with an index outside [0, L) dropped) -- bit for bit."""
    import ctypes
    import numpy as np
    from hvgan import lib
    B, h, w = 2, 32, 32
    L = h * w
    S = torch.randn(B, L, L, generator=torch.Generator().manual_seed(21 + adjoint))
    out = torch.empty(B, L, L, device='cuda')
    lib.get().call('hv_ca_fuse', lib.ptr(S.cuda()), lib.ptr(out), B, h, w, adjoint, lib.stream())
    torch.cuda.synchronize()
    Sn = S.numpy()
    idx = np.arange(L)
    tr = lambda a: (a % w) * h + a // w
    itr = lambda a: (a % h) * w + a // h
    ref = np.zeros((B, L, L), dtype=np.float32)

    def add(rows, cols, rok, cok):      # ref[:, p, l] += S[:, rows[p], cols[l]] where both are valid
        r = np.where(rok, rows, 0)
        c = np.where(cok, cols, 0)
        term = Sn[:, r][:, :, c] * (rok[:, None] & cok[None, :])[None].astype(np.float32)
        ref.__iadd__(term.astype(np.float32))

    if not adjoint:
        for d in (-1, 0, 1):
            a = tr(idx) + d
            ok_d = (a >= 0) & (a < L)
            base = itr(np.clip(a, 0, L - 1))
            for e in (-1, 0, 1):
                rows = base + e
                ok = ok_d & (rows >= 0) & (rows < L)
                add(np.clip(rows, 0, L - 1), np.clip(rows, 0, L - 1), ok, ok)
    else:
        for e in (-1, 0, 1):
            q = idx + e
            ok_e = (q >= 0) & (q < L)
            for d in (-1, 0, 1):
                a = tr(np.clip(q, 0, L - 1)) + d
                ok = ok_e & (a >= 0) & (a < L)
                rows = itr(np.clip(a, 0, L - 1))
                add(rows, rows, ok, ok)
    assert np.array_equal(out.cpu().numpy(), ref)


def test_score_fusion_adjoint_with_gs_and_coef_in_one_pass_matches_the_two_calls():
    """hv_ca_fuse_backward_prep (32 x 32 map: every workgroup owns a tile of dS0 and its mirror, dS0 never stored) against hv_ca_fuse(adjoint) followed by
    hv_ca_score_backward_prep: Gs bit for bit (same term order), coef to fp32 rounding (32 row blocks instead of 16 row chunks) and against the fp64 sum."""
    from hvgan import lib
    from hvgan.lib import ptr, stream
    B, h, w = 3, 32, 32
    L = h * w
    g = torch.Generator().manual_seed(77)
    dS1 = torch.randn(B, L, L, generator=g).cuda()
    S0 = torch.randn(B, L, L, generator=g).cuda()
    norm = (torch.rand(B, L, generator=g) + 0.5).cuda()
    norm[1, 17] = 5e-5                                       # a clamped norm: coef 0 there
    rnorm = (1.0 / norm.clamp_min(1e-4)).contiguous()
    Lc = lib.get()
    dS0 = torch.empty_like(dS1)
    Gs_ref = torch.empty_like(dS1)
    coef_ref = torch.zeros(17 * B, L, device='cuda')
    Lc.call('hv_ca_fuse', ptr(dS1), ptr(dS0), B, h, w, 1, stream())
    Lc.call('hv_ca_score_backward_prep', ptr(dS0), ptr(S0), ptr(norm), ptr(rnorm), ptr(Gs_ref), ptr(coef_ref), B, L, stream())
    Gs = torch.full_like(dS1, float('nan'))
    coef = torch.full((33 * B, L), float('nan'), device='cuda')
    Lc.call('hv_ca_fuse_backward_prep', ptr(dS1), ptr(S0), ptr(norm), ptr(rnorm), ptr(Gs), ptr(coef), B, h, w, stream())
    torch.cuda.synchronize()
    assert torch.equal(Gs, Gs_ref)
    c, cr = coef[:B].cpu(), coef_ref[:B].cpu()
    exact = -(dS0.double() * S0.double()).sum(1).cpu() / (norm.double().cpu() ** 2)
    exact[norm.cpu() <= 1e-4] = 0
    scale = exact.abs().max().item()
    assert c[1, 17] == 0
    assert (c.double() - exact).abs().max().item() <= 2e-6 * scale
    assert (c - cr).abs().max().item() <= 4e-6 * scale
    with pytest.raises(RuntimeError):                        # other maps keep the two calls
        Lc.call('hv_ca_fuse_backward_prep', ptr(dS1), ptr(S0), ptr(norm), ptr(rnorm), ptr(Gs), ptr(coef), 1, 16, 64, stream())


@pytest.mark.parametrize('B,R,C', [(2, 1024, 576), (3, 72, 40), (1, 64, 8), (2, 33, 50)])
def test_attention_operand_transposes_are_exact(B, R, C):
    """hv_transpose_batched_f16 (fp32 -> fp16) and hv_transpose_batched_h2h (fp16 -> fp16): the operand tables of the attention block's batched
    GEMMs.  64 x 64 tiles with 16-byte accesses when R and C are multiples of 8 (incl. ragged tile edges), the 32 x 32 element kernel otherwise;
    a conversion and a copy: exact against torch."""
    import hvgan  # noqa: F401
    from hvgan import lib
    from hvgan.lib import ptr, stream
    L = lib.get()
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(5)
    src = torch.randn(B, R, C, generator=g).to(dev)
    dst = torch.full((B, C, R), 7.0, dtype=torch.float16, device=dev)
    L.call('hv_transpose_batched_f16', ptr(src), ptr(dst), B, R, C, stream())
    assert torch.equal(dst, src.half().transpose(1, 2).contiguous())
    src_h = src.half().contiguous()
    dst2 = torch.full((B, C, R), 7.0, dtype=torch.float16, device=dev)
    L.call('hv_transpose_batched_h2h', ptr(src_h), ptr(dst2), B, R, C, stream())
    assert torch.equal(dst2, src_h.transpose(1, 2).contiguous())


@pytest.mark.parametrize('groups', [1, 2])
def test_batchnorm_backward_sums_from_the_data_gradient_epilogue(groups, monkeypatch):
    """fp16 mode, full-size PatchGAN (ndf 64): the batch-norm backward sums (sum g, sum g * xhat per channel and group) come out of the epilogue of the
    data gradient that writes g -- logits_dgrad_kernel (512 channels), conv_g4s1_kernel (256, stride 1) and conv_g4_kernel<1> (128, stride 2, four
    parity classes) -- instead of norm_reduce_kernel<1> (hv_conv_desc.bstats -> hv_norm_bwd_desc.partials).  Against the same backward with the
    reduction pass (HV_CONV_BSTATS=0): every parameter gradient and the input gradient agree to summation-order accuracy, for one statistics group and
    for the fake | real halves of a batched pass (two groups); a 72 x 56 map: partial tiles in both kernels."""
    monkeypatch.setenv('HV_PRECISION', 'fp16')
    from hvgan.models import networks
    from hvgan import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    net = networks.define_D(1, 64, 'basic', 3, 'batch', 'normal', 0.02, []).cuda()
    net.precision = 'fp16'
    net.train()
    x = torch.randn(4, 1, 72, 56, device=dev)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(networks, 'CONV_BSTATS', on)
        P = net.run_forward(x, training=True, groups=groups)
        dz = torch.randn(P.logits.shape, generator=torch.Generator().manual_seed(5)).to(dev) * 64.0
        dx = net.run_backward(P, dz, need_dx=True, param_grads=True)
        net.finish()
        torch.cuda.synchronize()
        used = [ent.get('bparts_used', 0) for ent in P.layers[1:-1]]
        assert all(u > 0 for u in used) if on else not any(used), (on, used)
        res[on] = ({k: p.grad.detach().clone() for k, p in net.named_parameters()}, dx.detach().clone())
    for k in res[True][0]:
        a, b = res[True][0][k], res[False][0][k]
        assert torch.isfinite(a).all() and b.abs().max().item() > 0, k
        assert (a - b).norm().item() <= 2e-3 * b.norm().item(), (k, (a - b).norm().item(), b.norm().item())
    assert (res[True][1] - res[False][1]).norm().item() <= 2e-3 * res[False][1].norm().item()
    # the same plan (same input size) with the other group count -- the batched fake | real pass and a plain pass of equal size share a plan in the train step
    monkeypatch.setattr(networks, 'CONV_BSTATS', True)
    P = net.run_forward(x, training=True, groups=3 - groups)
    dx2 = net.run_backward(P, torch.ones_like(P.logits), need_dx=True, param_grads=True)
    net.finish()
    torch.cuda.synchronize()
    assert all(ent.get('bparts_used', 0) > 0 for ent in P.layers[1:-1]) and torch.isfinite(dx2).all()


@pytest.mark.parametrize('size', [64, 128])
def test_attention_on_the_gram_route_against_the_cpu_oracle(size, monkeypatch):
    """The benchmarked attention route (fp16 mode: Gram-matrix scores, LDS-DMA batched GEMMs, fp16 attention matrix) against the ORACLE -- oracle.restate.
    contextual_attention, the CPU restatement pinned by fixture G2 -- at the shapes the fixture cannot reach: a 64-channel map of 64 x 64 (w = 32: the 256 x 256
    configuration) and 128 x 128 (w = 64, L = 4096: BASELINE config #5's slice size), two samples with DIFFERENT masks (sample 0's mask rules, reference
    models/inpaint_networks.py:314), output and input gradient.  fp16-mode gates of the G2 test."""
    from hvgan import engine, ops
    from oracle import restate as R
    monkeypatch.setenv('HV_PRECISION', 'fp16')
    dev = torch.device('cuda:0')
    B, C, H = 2, 64, size
    gen = torch.Generator().manual_seed(23)
    f = (torch.randn(B, C, H, H, generator=gen) * 0.5).half().float()          # (operands exactly representable in fp16: both sides see the same values)
    mask = torch.zeros(B, 1, 4 * H, 4 * H)
    mask[0, :, 4 * H // 3:4 * H // 3 + 40, :] = 1
    mask[1, :, :, 4 * H // 2:4 * H // 2 + 64] = 1
    coef = (torch.randn(B, C, H, H, generator=gen) * 0.05).half().float()
    # ---- oracle (CPU, fp32)
    torch.set_num_threads(max(1, min(32, (os.cpu_count() or 8))))
    fr = f.clone().requires_grad_(True)
    yr = R.contextual_attention(fr, fr, mask)
    (yr * coef).sum().backward()
    # ---- device
    plan = engine.AttentionPlan(B, H, H, C, dev, (4 * H, 4 * H))
    fa = ops.Act(f.permute(0, 2, 3, 1).contiguous().to(dev).half())
    out = ops.Act(torch.zeros(B, H, H, C, dtype=torch.float16, device=dev))
    plan.forward(fa, mask.to(dev), out, 'fp16')
    assert plan.gemm and plan.gram
    df = ops.Act(torch.zeros(B, H, H, C, dtype=torch.float16, device=dev))
    plan.backward(ops.Act(coef.permute(0, 2, 3, 1).contiguous().to(dev).half()), df, False, 'fp16')
    torch.cuda.synchronize()
    y = out.t.float().cpu().permute(0, 3, 1, 2)
    g = df.t.float().cpu().permute(0, 3, 1, 2)
    ys, gs = max(1.0, yr.abs().max().item()), max(1e-6, fr.grad.abs().max().item())
    assert _err(y, yr.detach()) <= 2e-2 * ys, (_err(y, yr.detach()), ys)
    rel = (g - fr.grad).norm().item() / fr.grad.norm().item()
    assert fr.grad.abs().max().item() > 0 and _err(g, fr.grad) <= 4e-2 * gs and rel <= 4e-2, (_err(g, fr.grad), gs, rel)


@pytest.mark.parametrize('size', [64, 128])
def test_attention_scores_on_the_pixel_gram_matrix_match_the_patch_table_route(size, monkeypatch):
    """fp16 mode, 64-channel map (the fine generator's attention input at 256 x 256 / 512 x 512 images: a 32- / 64-wide attention map): the matching scores
    and their gradient computed on the pixel Gram matrix (hv_ca_gram_scores: K = C per score, no patch tables; hv_ca_gram_backward: d fd = box(Gs) fd + the
    norm term) against the same block with the K = 9C patch GEMMs -- both round the same operands to fp16, only the summation order differs: scores,
    norms, output and the input gradient; the kernels taken are checked."""
    from hvgan import engine, ops
    monkeypatch.setenv('HV_PRECISION', 'fp16')
    dev = torch.device('cuda:0')
    B, C, H = 2, 64, size
    gen = torch.Generator().manual_seed(17)
    f = torch.randn(B, H, H, C, generator=gen).to(dev).half()
    mask = torch.zeros(B, 1, 4 * H, 4 * H, device=dev)
    mask[:, :, 4 * H // 3:4 * H // 3 + 40, :] = 1
    dout = (torch.randn(B, H, H, C, generator=gen) * 0.05).to(dev).half()
    res = {}
    for gram in (True, False):
        monkeypatch.setattr(engine, 'CA_GRAM', gram)
        plan = engine.AttentionPlan(B, H, H, C, dev, (4 * H, 4 * H))
        out = ops.Act(torch.zeros(B, H, H, C, dtype=torch.float16, device=dev))
        plan.forward(ops.Act(f), mask, out, 'fp16')
        assert plan.gemm and plan.gram == gram
        df = ops.Act(torch.zeros(B, H, H, C, dtype=torch.float16, device=dev))
        plan.backward(ops.Act(dout), df, False, 'fp16')
        torch.cuda.synchronize()
        res[gram] = dict(S0=plan.S0.t.float().cpu(), norm=plan.norm.cpu(), out=out.t.float().cpu(), df=df.t.float().cpu())
    a, b = res[True], res[False]
    assert (a['norm'] - b['norm']).abs().max().item() <= 1e-4 * b['norm'].abs().max().item()
    s = b['S0'].abs().max().item()
    assert s > 1.0 and (a['S0'] - b['S0']).abs().max().item() <= 2e-3 * s, ((a['S0'] - b['S0']).abs().max().item(), s)
    assert (a['out'] - b['out']).abs().max().item() <= 2e-2 * max(1.0, b['out'].abs().max().item())
    rel = (a['df'] - b['df']).norm().item() / b['df'].norm().item()
    assert b['df'].abs().max().item() > 0 and rel <= 2e-2, rel


def test_generator_with_other_widths_takes_the_materialised_concat(monkeypatch):
    """ngf = 32 in the fp16 mode: the [up-sampled | CAM channel] concat layers have 128 / 64 + 1 input channels, for which the extra-channel form of the
    filters-in-LDS kernel does not exist.  ConvNode.split_forward asks the C dispatch (hv_conv2d_supported) instead of mirroring its checks, so these
    layers read the materialised concat and the forward / backward run (round 3: RuntimeError 'unsupported'); attention with 128 channels keeps the
    patch-table route.  Outputs against the exact-fp32 mode on the same weights."""
    from hvgan.models.inpaint_networks import Generator
    from hvgan import synth
    dev = torch.device('cuda:0')
    b = synth.to_model_inputs(synth.make_batch(2, 256, seed=3))
    args = [b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev)]
    outs = {}
    for prec in ('fp16', 'fp32'):
        monkeypatch.setenv('HV_PRECISION', prec)
        torch.manual_seed(5)
        net = Generator({'input_dim': 1, 'ngf': 32}, True).cuda().train()
        net.precision = prec
        P = net.run_forward(*args, training=True)
        if prec == 'fp16':
            z = lambda t: torch.full_like(t, 1e-3)
            net.run_backward(P, z(P.coarse_seg), z(P.fine_seg), z(P.x_stage1), z(P.x_stage2), torch.zeros(2, 1, device=dev), torch.zeros(2, 1, device=dev))
            assert all(torch.isfinite(p.grad).all() for p in net.parameters() if p.grad is not None)
        torch.cuda.synchronize()
        outs[prec] = [t.detach().float().cpu().clone() for t in (P.coarse_seg, P.fine_seg, P.x_stage1, P.x_stage2)]
    for a, r in zip(outs['fp16'], outs['fp32']):
        assert torch.isfinite(a).all() and (a - r).abs().max().item() <= 2e-2, (a - r).abs().max().item()


def test_train_step_with_the_filters_in_lds_kernel_switched_off():
    """HV_CONV_LF=0 (a documented A/B knob, read once by the C side: own process): every 3x3 layer falls back to conv_halo2 -- incl. the concat layers, whose
    extra-channel form and pooled data gradient only the switched-off kernel serves: the Python side asks hv_conv2d_supported / pool2_ok of the dispatch
    and materialises the concat / the full-resolution gradient again.  Two train steps run and give the default build's losses."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import json, os, sys, torch
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
os.environ['HV_PRECISION'] = 'fp16'
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(1234)
m = Pix2PixModel(make_opt())
for s in range(2):
    m.set_input(synth.make_batch(2, 256, seed=1234 + s))
    m.optimize_parameters()
torch.cuda.synchronize()
print('LOSSES ' + json.dumps({k: float(v) for k, v in m.get_current_losses().items()}))
''' % (ROOT, ROOT)
    res = {}
    for lf in ('1', '0'):
        p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, HV_CONV_LF=lf), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        res[lf] = json.loads([l for l in p.stdout.decode().splitlines() if l.startswith('LOSSES ')][-1][7:])
    for k, v in res['1'].items():
        tol = 2e-2 if k in ('edge', 'D_real_2', 'D_fake_2') else 6e-3
        assert abs(res['0'][k] - v) <= tol * max(1.0, abs(v)), (k, res['0'][k], v)


def test_resident_filters_in_lds_kernel_is_bit_identical_to_the_per_tile_one():
    """conv_lfp_kernel (one workgroup per CU walking several tiles, filters by LDS-DMA, next patch prefetched; taken by the 64-input-channel and the
    extra-channel 3x3 layers when their grid is more than one round: 128 x 128 and 256 x 256 at bs 16) does the same arithmetic in the same order as
    conv_lf_kernel: one bs-16 generator forward + backward with HV_LF_PERSIST=1 / 0 (read once by the C side: own processes) gives bit-identical outputs and
    parameter gradients, and the resident kernel really ran in the first process."""
    import subprocess
    import sys
    from conftest import ROOT
    code = r'''
import hashlib, os, sys, torch
sys.path.insert(0, %r)
os.environ['HV_PRECISION'] = 'fp16'
import hvgan
from hvgan import synth, profiler, engine
from hvgan.models.inpaint_networks import Generator
torch.manual_seed(7)
dev = torch.device('cuda:0')
b = synth.to_model_inputs(synth.make_batch(16, 256, seed=11))
args = [b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev)]
G = Generator({'input_dim': 1, 'ngf': 16}, True).to(dev)
engine.SERIAL = True
prof = profiler.KernelTimer(); prof.enable()
P = G.run_forward(*args, training=True)
g = torch.Generator().manual_seed(3)
seeds = [torch.randn(t.shape, generator=g).to(dev) * 1e-2 for t in (P.coarse_seg, P.fine_seg, P.x_stage1, P.x_stage2)]
G.run_backward(P, seeds[0], seeds[1], seeds[2], seeds[3], None, None)
torch.cuda.synchronize()
names = sorted(set(prof.names.values())); prof.disable()
h = hashlib.sha256()
for t in [P.coarse_seg, P.fine_seg, P.x_stage1, P.x_stage2] + [q.grad for q in G.parameters() if q.grad is not None]:
    h.update(t.detach().float().cpu().numpy().tobytes())
print('HASH ' + h.hexdigest())
print('LFP ' + str(sum('conv_lfp_kernel' in n for n in names)))
''' % (ROOT,)
    res = {}
    for pz in ('1', '0'):
        p = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, HV_LF_PERSIST=pz), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
        assert p.returncode == 0, p.stdout.decode()[-3000:]
        out = p.stdout.decode().splitlines()
        res[pz] = ([l for l in out if l.startswith('HASH ')][-1], int([l for l in out if l.startswith('LFP ')][-1][4:]))
    assert res['1'][1] >= 3 and res['0'][1] == 0, res
    assert res['1'][0] == res['0'][0], res


@pytest.mark.parametrize('precision', ['fp16', 'fp32'])
def test_carried_slab_folds_are_bit_identical_to_folds_launched_on_their_own(precision, monkeypatch):
    """hv_wgrad_desc.pending / carry (ops.FoldChain): every weight gradient of a backward leaves its split-K slabs in one of two alternating buffers and the
    NEXT weight gradient of the stream takes their fold along -- as extra workgroups of its kernel (wgrad_tr_kernel, wgrad_halo_kernel) or, where that kernel
    has no room (the LDS-DMA and gather kernels), as a launch of its own; the last fold of a chain is launched by finish_backward / the side stream's owner.
    Same sums in the same order as the per-layer wgrad_reduce launches: generator and discriminator parameter gradients are BIT-identical with
    HV_FOLD_CHAIN=0 (every fold right behind its weight gradient), including an accumulating second backward (the split real / fake discriminator passes)."""
    monkeypatch.setenv('HV_PRECISION', precision)
    from hvgan import ops, synth
    from hvgan.models import networks
    from hvgan.models.inpaint_networks import Generator
    dev = torch.device('cuda:0')
    b = synth.to_model_inputs(synth.make_batch(2, 256, seed=3))
    args = [b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev)]
    x = torch.randn(2, 1, 256, 256, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for on in (True, False):
        monkeypatch.setattr(ops, 'FOLD_CHAIN', on)
        torch.manual_seed(5)
        net = Generator({'input_dim': 1, 'ngf': 16}, True).cuda().train()
        net.precision = precision
        P = net.run_forward(*args, training=True)
        z = lambda t: torch.full_like(t, 1e-2)
        net.run_backward(P, z(P.coarse_seg), z(P.fine_seg), z(P.x_stage1), z(P.x_stage2), torch.full((2, 1), 1e-2, device=dev), torch.full((2, 1), 1e-2, device=dev))
        torch.manual_seed(6)
        dnet = networks.define_D(1, 64, 'basic', 3, 'batch', 'normal', 0.02, []).cuda().train()
        dnet.precision = precision
        for acc in (False, True):       # second pass accumulates (split real / fake form)
            Pd = dnet.run_forward(x if not acc else -x, training=True)
            dnet.run_backward(Pd, torch.ones_like(Pd.logits), need_dx=False, param_grads=True, accumulate=acc)
        dnet.finish()
        torch.cuda.synchronize()
        assert all(c.pending is None for c in net.paramset().fold_chains.values()) and all(c.pending is None for c in dnet.paramset().fold_chains.values())
        res[on] = {('G', k): p.grad.detach().clone() for k, p in net.named_parameters()}
        res[on].update({('D', k): p.grad.detach().clone() for k, p in dnet.named_parameters()})
    for k in res[True]:
        assert torch.isfinite(res[True][k]).all() and res[False][k].abs().max().item() > 0, k
        assert torch.equal(res[True][k], res[False][k]), (k, (res[True][k] - res[False][k]).abs().max().item())




@pytest.mark.parametrize('norm,groups', [('batch', 2), ('batch', 1), ('instance', 1)])
def test_discriminator_head_normalisation_at_staging_is_bit_identical(norm, groups, monkeypatch):
    """fp16 mode: the last normalisation + LeakyReLU of the PatchGAN applied by the logits layer's kernel (hv_conv_desc.xn_*, HV_HEAD_NORM) against the
    separate normalisation pass: logits, every parameter gradient, the input gradient and the running statistics bit for bit."""
    from hvgan.models import networks
    from hvgan import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(3)
    ref = networks.define_D(1, 16, 'basic', 3, norm, 'normal', 0.02, [])
    sd = {k: v.clone() for k, v in ref.state_dict().items()}
    x = torch.randn(4, 1, 64, 64, generator=torch.Generator().manual_seed(4)).to(dev)
    res = []
    for on in (False, True):
        monkeypatch.setattr(networks, 'HEAD_NORM', on)
        net = networks.define_D(1, 16, 'basic', 3, norm, 'normal', 0.02, [])
        net.load_state_dict(sd)
        net.cuda().train()
        net.precision = 'fp16'
        P = net.run_forward(x, training=True, groups=groups)
        used = any(k[0] == 'head_xn' and v for ent in P.layers for k, v in ent.items() if isinstance(k, tuple))
        assert used == on
        loss = torch.zeros((), device=dev)
        dz = torch.empty_like(P.logits)
        ops.gan_loss(P.logits, True, 'vanilla', loss=loss, dz=dz)
        dx = net.run_backward(P, dz, need_dx=True, param_grads=True)
        net.finish()
        torch.cuda.synchronize()
        res.append((P.logits.clone(), dx.clone(), {k: p.grad.clone() for k, p in net.named_parameters()}, {k: v.clone() for k, v in net.state_dict().items()}))
    a, b = res
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        assert torch.equal(a[2][k], b[2][k]), k
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k
    assert a[0].abs().max().item() > 0 and a[1].abs().max().item() > 0
