"""Experiment: generator forward (+backward) on 16 slices at once vs two sub-batches of 8 (Infinity-Cache blocking)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('HV_PRECISION', 'fp16')
import hvgan
from hvgan import synth
from hvgan.models.inpaint_networks import Generator

torch.manual_seed(0)
net = Generator({'input_dim': 1, 'ngf': 16}, True).cuda().train()
dev = torch.device('cuda:0')


def inputs(B, seed):
    b = synth.to_model_inputs(synth.make_batch(B, 256, seed=seed))
    return [b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev)]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


full = inputs(16, 1)
halves = [inputs(8, 2), inputs(8, 3)]
quarters = [inputs(4, 4 + i) for i in range(4)]


def fwd(sets):
    return [net.run_forward(*s, training=True) for s in sets]


def fwd_bwd(sets):
    for s in sets:
        P = net.run_forward(*s, training=True)
        z = lambda t: torch.ones_like(t) * 1e-3
        B = s[0].shape[0]
        net.run_backward(P, z(P.coarse_seg), z(P.fine_seg), z(P.x_stage1), z(P.x_stage2), torch.zeros(B, 1, device=dev), torch.zeros(B, 1, device=dev))


for name, fn in (('forward', fwd), ('forward+backward', fwd_bwd)):
    print('%-17s  1x16: %.2f ms   2x8: %.2f ms   4x4: %.2f ms' % (name, timeit(lambda: fn([full])), timeit(lambda: fn(halves)), timeit(lambda: fn(quarters))))


def graphed(fn):
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    return g.replay


from hvgan import engine
engine.SERIAL = True
for name, fn in (('forward', fwd), ('forward+backward', fwd_bwd)):
    r = [graphed(lambda s=s: fn(s)) for s in ([full], halves, quarters)]
    print('graph %-17s  1x16: %.2f ms   2x8: %.2f ms   4x4: %.2f ms' % (name, timeit(r[0], 20), timeit(r[1], 20), timeit(r[2], 20)))
