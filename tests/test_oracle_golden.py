"""The CPU oracle (oracle/restate.py) against the golden vectors the reference produced
(oracle/make_golden.py).  CPU only; this is what pins the oracle."""
import os

import torch

from oracle import restate as R
from conftest import load_golden

TOL = 2e-5


def close(a, b, tol=TOL):
    a, b = a.double(), b.double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs().max().item()
    assert err <= tol * max(1.0, b.abs().max().item()), err


def test_g1_generator_train_eval_and_grads():
    g = load_golden('g1_generator_mini')
    sd = {k: v.clone() for k, v in g['sd'].items()}
    params = [k for k in sd if k.endswith('weight_orig') or k.endswith('.bias') or k.endswith('fc_height.weight')]
    for k in params:
        sd[k].requires_grad_(True)
    outs, upd = R.generator_forward(sd, g['x'], g['mask'], g['cam'], g['ratio'], training=True)
    names = ('coarse_seg', 'fine_seg', 'x_stage1', 'x_stage2', 'pred1_h', 'pred2_h')
    for n, o in zip(names, outs):
        close(o.detach(), g['train'][n])
    loss = sum((o * g['coef'][str(i)]).sum() for i, o in enumerate(outs))
    close(loss.detach(), g['loss'], 1e-5)
    loss.backward()
    for k in params:
        ref = g['grads'][k]
        err = (sd[k].grad - ref).abs().max().item()
        assert err <= 2e-4 * max(1.0, ref.abs().max().item()), (k, err)
    for k, v in upd.items():
        close(v, g['bufs_after'][k], 1e-5)
    sd2 = {k: v.detach().clone() for k, v in sd.items()}
    sd2.update({k: v for k, v in g['bufs_after'].items()})
    with torch.no_grad():
        oute, upde = R.generator_forward(sd2, g['x'], g['mask'], g['cam'], g['ratio'], training=False)
    assert not upde
    for n, o in zip(names, oute):
        close(o, g['eval'][n])


def test_g2_attention_batch0_mask_quirk_and_grad():
    g = load_golden('g2_attention')
    f = g['f'].clone().requires_grad_(True)
    y = R.contextual_attention(f, f, g['mask'])
    close(y.detach(), g['y'])
    (y * g['coef']).sum().backward()
    close(f.grad, g['grad_f'], 1e-4)


def test_g3_discriminator():
    """'basic' (three layers) and, G3n, define_D('n_layers', n_layers_D in {2, 4}) (reference models/networks.py:198-199)."""
    for name, norm in (('g3_disc_batch', 'batch'), ('g3_disc_instance', 'instance'), ('g3n_disc_n2_batch', 'batch'), ('g3n_disc_n4_batch', 'batch'),
                       ('g3n_disc_n4_instance', 'instance')):
        g = load_golden(name)
        sd = {k: v.clone() for k, v in g['sd'].items()}
        params = [k for k in sd if k.endswith('.weight') or k.endswith('.bias')]
        for k in params:
            sd[k].requires_grad_(True)
        x0 = g['x']['0'].clone().requires_grad_(True)
        y0, upd = R.disc_forward(sd, x0, norm, True)
        close(y0.detach(), g['y']['0'])
        loss = R.gan_loss(y0, True)
        close(loss.detach(), g['loss'])
        loss.backward()
        for k in params:
            close(sd[k].grad, g['grads'][k], 1e-4)
        close(x0.grad, g['grad_x'], 1e-4)
        with torch.no_grad():
            for k, v in upd.items():
                sd[k] = v
            for i in (1, 2):
                y, upd = R.disc_forward(sd, g['x'][str(i)], norm, True)
                close(y, g['y'][str(i)])
                for k, v in upd.items():
                    sd[k] = v
            for k, v in g.get('sd_after', {}).items():
                close(sd[k].double(), v.double(), 1e-5)
            ye, _ = R.disc_forward(sd, g['x']['0'], norm, False)
            close(ye, g['y_eval'])


def test_g4_small_ops():
    g = load_golden('g4_small_ops')
    close(R.sobel(g['m']), g['sobel_m'])
    close(R.sobel(g['soft']), g['sobel_soft'])
    close(R.dice_coeff(g['soft'], g['m']), g['dice'])
    for mode in ('vanilla', 'lsgan'):
        close(R.gan_loss(g['pred'], True, mode), g['gan_%s_real' % mode])
        close(R.gan_loss(g['pred'], False, mode), g['gan_%s_fake' % mode])


def test_g6_unet():
    g = load_golden('g6_unet_mini')
    sd = {k: v.clone() for k, v in g['sd'].items()}
    params = [k for k in sd if (k.endswith('.weight') or k.endswith('.bias'))]
    for k in params:
        sd[k].requires_grad_(True)
    (ct, mk), upd = R.unet_forward(sd, g['x'], 5, True)
    close(ct.detach(), g['ct'])
    close(mk.detach(), g['mk'])
    loss = (ct - g['tgt']).abs().mean() + (mk * g['tgt']).mean()
    close(loss.detach(), g['loss'])
    loss.backward()
    for k in params:
        close(sd[k].grad, g['grads'][k], 2e-4)
    with torch.no_grad():
        for k, v in upd.items():
            sd[k] = v
        for k, v in g['sd_after'].items():
            close(sd[k].double(), v.double(), 1e-5)
        (cte, mke), _ = R.unet_forward(sd, g['x'], 5, False)
        close(cte, g['ct_eval'])
        close(mke, g['mk_eval'])


def test_g6b_unet_dropout_quirk():
    """use_dropout=True hands nn.Dropout the boolean (p = 1.0): zeros in train mode, identity in eval mode (reference models/UnetG_CT_mask.py:73-78)."""
    g = load_golden('g6b_unet_dropout')
    sd = {k: v.clone() for k, v in g['sd'].items()}
    with torch.no_grad():
        (ct, mk), upd = R.unet_forward(sd, g['x'], 5, True, use_dropout=True)
        close(ct, g['ct'])
        close(mk, g['mk'])
        for k, v in upd.items():
            sd[k] = v
        for k, v in g['sd_after'].items():
            close(sd[k].double(), v.double(), 1e-5)
        (cte, mke), _ = R.unet_forward(sd, g['x'], 5, False, use_dropout=True)
        close(cte, g['ct_eval'])
        close(mke, g['mk_eval'])


def test_g8_rhlv_oracle_matches_reference_outputs():
    """RHLV restatement (oracle.restate.rhlv / rhlv_volume) against the reference's calculate_rhlv / calculate_heights outputs (G8)."""
    import numpy as np
    from oracle import restate as R
    g = load_golden('g8_rhlv')
    names = sorted({k.split('/')[0] for k in g.keys()})
    assert len(names) >= 6
    for n in names:
        idx, div, thr, cz, length = (float(v) for v in g[n + '/params'])
        fake, label = np.asarray(g[n + '/fake'], dtype=np.float64), np.asarray(g[n + '/label'], dtype=np.float64)
        res, means = R.rhlv_volume(fake, label, idx, int(div), thr)
        assert np.array_equal(np.array(res), np.asarray(g[n + '/out'])), (n, res)          # same numpy operations: bit-identical
        assert np.array_equal(np.array(means), np.asarray(g[n + '/means'])), n
        sf, sl = (fake == idx).astype(np.float64), (label == idx).astype(np.float64)
        res2, _ = R.rhlv(sf, sl, int(cz), int(length), thr)
        assert np.array_equal(np.array(res2), np.asarray(g[n + '/out']))


G9_VOLUME_KW = {1: dict(), 2: dict(), 3: dict(H=112, W=72, Z=14)}


def g9_cases():
    """(name, float64 volumes, vertebra id, normal list, numpy seed, expected dict) for every G9 case."""
    import numpy as np
    import hvgan  # noqa: F401
    from hvgan import synth
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g9_assemble.npz'))
    for name in z['cases']:
        name = str(name)
        vseed, vert_id, nseed, sl, height, x1, x2, h2 = (int(v) for v in z[name + '/meta'])
        ct, label, cam = synth.make_spine_volume(vseed, **G9_VOLUME_KW[vseed])
        exp = {k: z[name + '/' + k] for k in ('A', 'B', 'A_mask', 'mask', 'normal_vert', 'CAM')}
        exp.update(slice=sl, height=height, x1=x1, x2=x2, h2=h2, slice_ratio=float(z[name + '/slice_ratio'][0]))
        yield name, ct, label, cam, vert_id, [str(v) for v in z[name + '/normals']], nseed, exp


def test_g9_batch_assembly_oracle_matches_reference_getitem():
    """oracle.restate.dataset_item_u8 (slice draw, component filter, row re-stacking, uint8 quantisation) against the outputs of the
    reference's AlignedDataset.__getitem__ (G9): bit-identical images and metadata."""
    import numpy as np
    n = 0
    for name, ct, label, cam, vert_id, normals, nseed, exp in g9_cases():
        np.random.seed(nseed)
        got = R.dataset_item_u8(ct.astype(np.float64), label.astype(np.float64), cam.astype(np.float64) * 255, vert_id, normals)
        for k in ('slice', 'height', 'x1', 'x2', 'h2'):
            assert got[k] == exp[k], (name, k, got[k], exp[k])
        assert got['slice_ratio'] == exp['slice_ratio']
        for k in ('A', 'B', 'A_mask', 'mask', 'normal_vert', 'CAM'):
            assert np.array_equal(got[k], exp[k]), (name, k)
        n += 1
    assert n >= 6


def g10_cases():
    """(name, ct, label, cam slices, vertebra id, ratio tensor, recording model, expected arrays or None) for every G10 case."""
    import numpy as np
    from oracle.make_golden_infer import CASES, make_slice, RecordingModel
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g10_infer.npz'))
    for i, (name, rows, vert_id, frac, specks) in enumerate(CASES):
        ct, label, cam = make_slice(i, rows, vert_id, specks)
        exp = None if (name + '/none') in z.files else {k: z[name + '/' + k] for k in ('in_ct', 'in_mask', 'in_cam', 'label_fake', 'ct_fake', 'height')}
        yield name, ct, label, cam, vert_id, torch.tensor([0.125 * i]), RecordingModel(500 + i, frac), exp


def test_g10_inference_slice_oracle_matches_reference_run_model():
    """oracle.restate.infer_run_model (component filter, bbox / too-tall window, 41-row mask band, re-stacking, uint8 quantisation, then the
    re-compositing around the network's outputs) against the reference's run_model driven by the same recording stand-in network (G10)."""
    import numpy as np
    n = 0
    for name, ct, label, cam, vert_id, ratio, model, exp in g10_cases():
        res = R.infer_run_model(model, cam.copy(), label.copy(), ct.copy(), vert_id, ratio)
        if exp is None:
            assert res is None, name
            continue
        lab, ctf, height = res
        ctb, mb, cinv, r = model.inputs
        assert torch.equal(ctb[0], R.to_tensor_u8(exp['in_ct'], True)) and torch.equal(mb[0], R.to_tensor_u8(exp['in_mask'], False)), name
        assert torch.equal(cinv[0], 1 - R.to_tensor_u8(exp['in_cam'], False)) and torch.equal(r, ratio), name
        assert height == int(exp['height'][0])
        assert np.array_equal(np.asarray(lab, dtype=np.float64), exp['label_fake']), name
        assert np.array_equal(np.asarray(ctf, dtype=np.float64), exp['ct_fake']), name
        n += 1
    assert n >= 6


def g11_cases():
    """(name, model-attribute batch, fake generator outputs, expected dict) for fixture G11 (oracle/make_golden_eval.py)."""
    import numpy as np
    import hvgan  # noqa: F401
    from hvgan import synth
    from oracle.make_golden_eval import fake_generator_outputs
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'g11_eval.npz'))
    for name, (B, bseed, oseed) in zip(g['cases'], g['params']):
        name = str(name)
        b = synth.to_model_inputs(synth.make_batch(int(B), 256, seed=int(bseed)))
        outs = fake_generator_outputs(b, int(oseed), name.endswith('tall'))
        exp = {k[len(name) + 1:]: g[k] for k in g.files if k.startswith(name + '/')}
        yield name, b, outs, exp


def test_g11_eval_metrics_oracle_matches_reference_evaluate_model():
    """oracle.restate.eval_sample_metrics == the reference's evaluate_model (train.py:50-160) on fixture G11: Dice, IoU and the height
    error are the reference's own arithmetic; for SSIM / PSNR (skimage: restated, parity unpinned) the fixture pins what the reference
    feeds them -- the masked, composited operands (checksums + sparse samples) and both data_range values."""
    import numpy as np
    for name, b, outs, exp in g11_cases():
        coarse, fine, _, stage2, _, _, pred2 = outs
        m, inp = R.eval_sample_metrics(stage2, fine, coarse, pred2, b)
        assert np.allclose(m[:, 2:], exp['per_sample'][:, 2:], rtol=1e-6, atol=1e-7), name        # dice, iou, diff_h: reference code
        assert np.allclose(m[:, :2], exp['per_sample'][:, :2], rtol=1e-9), name                   # same restated functions both sides
        assert np.allclose(m.mean(0), exp['avg'], rtol=1e-5), name
        for i in range(inp.shape[0]):
            gt, mk, x = b['real_B'][i].numpy(), b['mask'][i].numpy(), inp[i].numpy()
            a, y = (gt * mk).squeeze(), (x * mk).squeeze()
            ops = exp['operands'][i]
            assert abs(a.astype(np.float64).sum() - ops[0]) <= 1e-9 * max(1, abs(ops[0])) and abs(y.astype(np.float64).sum() - ops[1]) <= 1e-9 * max(1, abs(ops[1]))
            assert abs(float(x.max() - x.min()) - ops[3]) <= 1e-7 and abs(float(x.max() - gt.min()) - ops[4]) <= 1e-7
            assert np.array_equal(y[::4, ::4], exp['masked_result_sparse/%d' % i]), (name, i)
    # a hand case for the SSIM restatement: identical images give 1, and the published closed form on constant images
    a = np.full((16, 16), 0.25, np.float32)
    assert abs(R.eval_ssim(a, a, 1.0) - 1.0) < 1e-7
    c = np.full((16, 16), 0.75, np.float32)
    want = (2 * 0.25 * 0.75 + 1e-4) / (0.25 ** 2 + 0.75 ** 2 + 1e-4)       # variances vanish: S = luminance term
    assert abs(R.eval_ssim(a, c, 1.0) - want) < 1e-6
