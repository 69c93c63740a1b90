"""Post-generator operators against the reference's own outputs on hand-built inputs at the decision boundaries (fixture G12,
oracle/make_golden_shrm.py: the reference's Pix2PixModel.forward behind a stand-in generator, models/pix2pix_model.py:180-264), and the small
operators against fixture G4 (Sobel incl. the clip-to-1 branch, diceCoeff, GANLoss vanilla / lsgan): SURVEY section 8c."""
import ctypes
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _inputs():
    from oracle.make_golden_shrm import build_inputs
    return build_inputs()


def test_g12_fixture_inputs_are_reproducible():
    """The hand-built inputs are formulas; the fixture keeps every 4th column of them as a cross-check."""
    g = np.load(os.path.join(GOLD, 'g12_shrm.npz'))
    I = _inputs()
    for k, v in I.items():
        ref = g['in::' + k]
        got = v.numpy() if v.dim() < 4 else v[..., ::4].numpy()
        assert np.array_equal(got, ref), k
    # the cases the fixture is there for
    p2 = g['res::pred2_h'][0]
    assert np.ceil(p2[0]) == 30 and np.ceil(p2[1]) == 31 and p2[2] < I['height'][2].item() and p2[3] == 28.0


@pytest.mark.gpu
def test_post_generator_matches_reference_forward_on_boundary_cases():
    import hvgan
    from hvgan import lib, ops
    from hvgan.lib import ptr, stream
    g = np.load(os.path.join(GOLD, 'g12_shrm.npz'))
    I = _inputs()
    dev = torch.device('cuda:0')
    B, H, W = 8, 256, 256
    T = {k: v.to(dev).contiguous() for k, v in I.items()}
    L = lib.get()
    d = L.hv_postg_desc()
    outs = {n: torch.full((B, 1, H, W), 7.0, device=dev) for n in ('fake_B', 'fake_B_coarse', 'fake_B_local', 'real_B_local', 'fine_bin', 'coarse_bin')}
    p1h, p2h = torch.zeros(1, B, device=dev), torch.zeros(1, B, device=dev)
    rows = torch.zeros(B, 4, dtype=torch.int32, device=dev)
    for f, t in (('real_B', T['real_B']), ('mask', T['mask']), ('x_stage1', T['x_stage1']), ('x_stage2', T['x_stage2']), ('fine_seg', T['fine']),
                 ('coarse_seg', T['coarse']), ('pred1', T['pred1']), ('pred2', T['pred2']), ('height', T['height']), ('x1', T['x1']), ('x2', T['x2']),
                 ('maxheight', T['maxheight']), ('pred1_h', p1h), ('pred2_h', p2h), ('rows', rows)) + tuple(outs.items()):
        setattr(d, f, ptr(t).value)
    d.B, d.H, d.W, d.half_band = B, H, W, 35
    L.call('hv_post_generator', ctypes.byref(d), stream())
    torch.cuda.synchronize()
    names = dict(fake_B='fake_B', fake_B_coarse='fake_B_coarse', fake_B_local='fake_B_local', real_B_local='real_B_local',
                 fine_bin='fake_B_mask_raw', coarse_bin='coarse_seg_binary')
    for k, rk in names.items():
        got = outs[k].cpu().numpy()[..., ::4]
        assert np.array_equal(got, g['res::' + rk]), (k, np.abs(got - g['res::' + rk]).max())       # bit-exact: copies, products with 0/1, thresholds
    assert np.array_equal(p1h.cpu().numpy(), g['res::pred1_h']) and np.array_equal(p2h.cpu().numpy(), g['res::pred2_h'])
    # row bounds the kernel derived on the device == the reference's x_upper / x_bottom (read back from the composited image)
    r = rows.cpu().numpy()
    for i in range(B):
        h2 = max(int(np.ceil(g['res::pred2_h'][0, i])), int(I['height'][i]))
        xu = int(I['x1'][i]) - (h2 - int(I['height'][i])) // 2
        assert (r[i, 0], r[i, 1]) == (xu, xu + h2), (i, r[i])
    # generic compositing entry point (evaluation / inference callers)
    out = torch.zeros(B, 1, H, W, device=dev)
    rows2 = torch.zeros(B, 2, dtype=torch.int32, device=dev)
    L.call('hv_shrm_composite', ptr(T['x_stage2']), ptr(T['real_B']), ptr(p2h.view(-1).contiguous()), ptr(T['height']), ptr(T['x1']), ptr(T['x2']),
           ptr(out), ptr(rows2), B, H, W, stream())
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy()[..., ::4], g['res::fake_B'])
    # both Sobel edge maps (binary inputs: the clip-to-1 branch is taken at most edge pixels)
    for src, rk in ((T['real_B_mask'], 'real_edges'), (outs['fine_bin'], 'fake_edges')):
        e = ops.sobel(src.contiguous())
        torch.cuda.synchronize()
        assert np.abs(e.cpu().numpy()[..., ::4] - g['res::' + rk]).max() <= 1e-6, rk


@pytest.mark.gpu
def test_small_operators_match_reference_g4():
    """hv_sobel (hard and soft inputs), the generator-loss kernel's Dice term and hv_gan_loss (vanilla, lsgan; real / fake) against the
    reference's Sobel / diceCoeff / GANLoss (fixture G4, oracle/make_golden.py:141-154)."""
    import hvgan
    from hvgan import lib, ops
    from hvgan.lib import ptr, stream
    g = np.load(os.path.join(GOLD, 'g4_small_ops.npz'))
    dev = torch.device('cuda:0')
    m, soft = torch.from_numpy(g['m']).to(dev), torch.from_numpy(g['soft']).to(dev)
    assert np.abs(ops.sobel(m).cpu().numpy() - g['sobel_m']).max() <= 1e-6
    s = ops.sobel(soft).cpu().numpy()
    assert np.abs(s - g['sobel_soft']).max() <= 2e-6 and (g['sobel_soft'] == 1.0).any() and (g['sobel_soft'] < 1.0).any()       # both branches of the clip
    pred = torch.from_numpy(g['pred']).to(dev)
    for mode in ('vanilla', 'lsgan'):
        for real in (True, False):
            loss = torch.zeros((), device=dev)
            dz = torch.zeros_like(pred)
            ops.gan_loss(pred.contiguous(), real, mode, loss=loss, dz=dz)
            ref = float(g['gan_%s_%s' % (mode, 'real' if real else 'fake')])
            assert abs(loss.item() - ref) <= 1e-6 * max(1.0, abs(ref)), (mode, real, loss.item(), ref)
            # gradient of the mean loss wrt the logits
            z = torch.from_numpy(g['pred']).requires_grad_(True)
            t = torch.ones_like(z) if real else torch.zeros_like(z)
            (torch.nn.functional.binary_cross_entropy_with_logits(z, t) if mode == 'vanilla' else torch.nn.functional.mse_loss(z, t)).backward()
            assert (dz.cpu() - z.grad).abs().max().item() <= 1e-7
    # Dice through the generator-loss kernel: G_Dice = (1 - dice(fine_seg, real_B_mask)) * 15, coarse_Dice = (1 - dice(coarse_seg, normal_vert)) * 10
    L = lib.get()
    B, _, H, W = g['m'].shape
    d = L.hv_gloss_desc()
    z = lambda *s_: torch.zeros(*s_, device=dev)
    img = z(B, 1, H, W)
    bufs = dict(fake_B=img, fake_B_coarse=img, real_B=img, mask=torch.ones(B, 1, H, W, device=dev), fine_seg=soft.contiguous(), coarse_seg=soft.contiguous(),
                real_B_mask=m.contiguous(), normal_vert=m.contiguous(), fake_edges=img, real_edges=img, pred1_h=torch.full((1, B), 20.0, device=dev),
                pred2_h=torch.full((1, B), 20.0, device=dev), height=torch.full((B,), 20, dtype=torch.int64, device=dev),
                maxheight=torch.full((B,), 40, dtype=torch.int64, device=dev), losses=z(8), d_fake_B=z(B, 1, H, W), d_fake_B_coarse=z(B, 1, H, W),
                d_fine_seg=z(B, 1, H, W), d_coarse_seg=z(B, 1, H, W), d_pred1=z(B, 1), d_pred2=z(B, 1))
    for f, t in bufs.items():
        setattr(d, f, ptr(t).value)
    d.lambda_L1, d.B, d.H, d.W, d.grad_scale = 200.0, B, H, W, 0.0
    ws = torch.zeros(max(1, L.size('hv_generator_losses_workspace_bytes', B)) // 4 + 4, device=dev)
    d.workspace, d.workspace_bytes = ptr(ws).value, ws.numel() * 4
    L.call('hv_generator_losses', ctypes.byref(d), stream())
    torch.cuda.synchronize()
    dice = float(g['dice'])
    lo = bufs['losses'].cpu().numpy()
    assert abs(lo[1] - (1 - dice) * 15) <= 2e-5 and abs(lo[2] - (1 - dice) * 10) <= 2e-5, (lo, dice)
