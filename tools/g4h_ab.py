"""Bit-identity A/B of the half-width 4x4 kernels (conv_g4h_kernel, HV_G4H) against the eight-wave ones, on the full-size PatchGAN in the fp16 mode:
    HV_G4H=0  python tools/g4h_ab.py run /tmp/a.pt
    HV_G4H=15 python tools/g4h_ab.py run /tmp/b.pt
    python tools/g4h_ab.py cmp /tmp/a.pt /tmp/b.pt
(the dispatch knobs are read once per process, hence two processes).  `run` also prints the time of the discriminator pass alone."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(dst):
    os.environ['HV_PRECISION'] = 'fp16'
    import hvgan  # noqa: F401
    from hvgan.models import networks
    dev = torch.device('cuda:0')
    out = {}
    for tag, B, H, W, groups in (('b32', 32, 256, 256, 2), ('b16', 16, 256, 256, 1), ('ragged', 4, 72, 56, 2), ('b16x128', 16, 128, 128, 1)):
        torch.manual_seed(3)
        net = networks.define_D(1, 64, 'basic', 3, 'batch', 'normal', 0.02, []).cuda()
        net.precision = 'fp16'
        net.train()
        x = torch.randn(B, 1, H, W, generator=torch.Generator().manual_seed(7)).to(dev)
        P = net.run_forward(x, training=True, groups=groups)
        dz = (torch.randn(P.logits.shape, generator=torch.Generator().manual_seed(5)) * 64.0).to(dev)
        dx = net.run_backward(P, dz, need_dx=True, param_grads=True)
        net.finish()
        torch.cuda.synchronize()
        out[tag] = {'logits': P.logits.detach().cpu().clone(), 'dx': dx.detach().cpu().clone(),
                    'grads': {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()},
                    'bufs': {k: v.detach().cpu().clone() for k, v in net.named_buffers()}}
        if tag in ('b32', 'b16'):
            def step():
                P = net.run_forward(x, training=True, groups=groups)
                net.run_backward(P, dz, need_dx=True, param_grads=True)
                net.finish()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(5):
                    step()
            g.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); g.replay(); e1.record()
            torch.cuda.synchronize()
            print('HV_G4H=%s %s: discriminator forward + backward %.1f us' % (os.environ.get('HV_G4H', 'default'), tag, e0.elapsed_time(e1) * 100.0), flush=True)
    torch.save(out, dst)


def cmp(a, b):
    A, B = torch.load(a), torch.load(b)
    bad = 0

    def walk(x, y, path):
        nonlocal bad
        if isinstance(x, dict):
            for k in x:
                walk(x[k], y[k], path + '/' + str(k))
        elif not torch.equal(x, y):
            bad += 1
            print('DIFF %s  max|d| %.3e of %.3e' % (path, (x.double() - y.double()).abs().max().item(), x.double().abs().max().item()))
    walk(A, B, '')
    print('bit-identical' if not bad else '%d tensors differ' % bad)
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(run(sys.argv[2]) if sys.argv[1] == 'run' else cmp(sys.argv[2], sys.argv[3]))
