"""What one launch per layer for all three discriminators would buy: three PatchGANs (batch 2B each, fake | real groups) forward + backward on three streams inside one
hipGraph -- what the step does -- against ONE PatchGAN at batch 6B (six groups) on one stream: the same work per kernel family with a third of the launches (the weights are
shared in the probe, which a grouped kernel would index per discriminator: same bytes, same FLOPs).
    python tools/d_group_probe.py [B=16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HV_PRECISION', 'fp16')
import torch

import hvgan  # noqa: F401
from hvgan import engine
from hvgan.models import networks

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')


def make(nb, groups):
    torch.manual_seed(3)
    net = networks.define_D(1, 64, 'basic', 3, 'batch', 'normal', 0.02, []).cuda()
    net.precision = 'fp16'
    net.train()
    x = torch.randn(nb, 1, 256, 256, generator=torch.Generator().manual_seed(7)).to(dev)
    P = net.run_forward(x, training=True, groups=groups)
    dz = (torch.randn(P.logits.shape, generator=torch.Generator().manual_seed(5)) * 64.0).to(dev)

    def step(need_dx):
        P = net.run_forward(x, training=True, groups=groups)
        net.run_backward(P, dz, need_dx=need_dx, param_grads=True)
        net.finish()
    return step


def timed(fn, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


for need_dx, what in ((False, 'discriminator update (2B per net, fake | real, no dx)'), (True, 'generator part (dx wanted)')):
    nb = 2 * B if not need_dx else B
    groups = 2 if not need_dx else 1
    three = [make(nb, groups) for _ in range(3)]
    streams = [torch.cuda.Stream(dev) for _ in range(3)]

    def concurrent():
        main = torch.cuda.current_stream(dev)
        for st, f in zip(streams, three):
            st.wait_stream(main)
            with torch.cuda.stream(st):
                f(need_dx)
        for st in streams:
            main.wait_stream(st)

    def serial():
        for f in three:
            f(need_dx)

    one = make(3 * nb, 3 * groups)
    t3 = timed(concurrent)
    ts = timed(serial)
    t1 = timed(lambda: one(need_dx))
    print('%s: three nets on three streams %.0f us, one after the other %.0f us, one net at 3x the batch %.0f us' % (what, t3, ts, t1), flush=True)
