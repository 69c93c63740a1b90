"""Generator / ContextualAttention / discriminator on the HIP path (fp32 mode) against the golden vectors the
reference produced, at the north-star tolerance |d| <= 1e-3 (activations) -- observed errors are ~1e-5."""
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _err(a, b):
    return (a.detach().cpu().double() - b.double()).abs().max().item()


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


def test_g1_generator_module_call_and_autograd_bridge():
    g = load_golden('g1_generator_mini')
    net = _gen(g)
    net.train()
    dev = torch.device('cuda:0')
    x, mask, cam, ratio = (g[k].to(dev) for k in ('x', 'mask', 'cam', 'ratio'))
    o = net(x, mask, cam, ratio)
    assert len(o) == 7 and o[4].shape == (2, 3, 64, 64)
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
    (y * g['coef'].to(dev)).sum().backward()
    assert _err(f.grad, g['grad_f']) <= TOL * max(1.0, g['grad_f'].abs().max().item())


@pytest.mark.parametrize('norm', ['batch', 'instance'])
def test_g3_discriminator(norm):
    from hvgan.models import networks
    g = load_golden('g3_disc_%s' % norm)
    dev = torch.device('cuda:0')
    net = networks.define_D(1, 8, 'basic', 3, norm, 'normal', 0.02, [])
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
