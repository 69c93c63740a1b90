"""Experiment: run the discriminators' real-image pass concurrently with the generator forward (timing only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ.setdefault('HV_PRECISION', 'fp16')
import hvgan
from hvgan import synth, ops
from hvgan.models.pix2pix_model import Pix2PixModel
torch.manual_seed(1234)
opt = bench.make_opt('fp16')
m = Pix2PixModel(opt); m.setup(opt)
m.use_graph = False
m.set_input(synth.make_batch(16, 256, seed=1234))
for _ in range(3):
    m.optimize_parameters()
torch.cuda.synchronize()
reals = {1: m.real_B, 2: m.real_B_mask, 3: m.real_B_local.clone()}
mode = m.opt.gan_mode


def step_overlap():
    main = torch.cuda.current_stream()
    for k in (1, 2, 3):
        side = m._d_streams[k - 1]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            net = getattr(m, 'netD_%d' % k)
            P = net.run_forward(reals[k], training=True, prep=True)
            dz = m._buf('dz%d' % k, P.logits)
            ops.gan_loss(P.logits, True, mode, loss=m._loss_slot(2 * k + 1), dz=dz, grad_weight=0.5)
            net.run_backward(P, dz, need_dx=False, param_grads=True, accumulate=False)
    m.forward()
    fakes = {1: m.fake_B, 2: m.fake_B_mask_raw, 3: m.fake_B_local}
    for k in (1, 2, 3):
        side = m._d_streams[k - 1]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            net = getattr(m, 'netD_%d' % k)
            P = net.run_forward(fakes[k], training=True, prep=False)
            dz = m._buf('dz%d' % k, P.logits)
            ops.gan_loss(P.logits, False, mode, loss=m._loss_slot(2 * k), dz=dz, grad_weight=0.5)
            net.run_backward(P, dz, need_dx=False, param_grads=True, accumulate=True)
            net.finish()
            getattr(m, 'optimizer_D_%d' % k).step(sync_lr=False)
            m._g_step_D(k)
    m._join_d(main)
    m.backward_G(d_done=True)
    m.optimizer_G.step(sync_lr=False)


def timeit(fn, n=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for _ in range(2):
    step_overlap()
print('baseline eager step  %.2f ms' % timeit(m.optimize_parameters))
print('real pass overlapped %.2f ms' % timeit(step_overlap))
print('baseline eager step  %.2f ms' % timeit(m.optimize_parameters))
print('real pass overlapped %.2f ms' % timeit(step_overlap))

# the same two step orders as captured hipGraphs
def capture(fn):
    s = torch.cuda.Stream()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        fn()
    return g


def base_step():
    m._phase_a(); m._phase_b(); m._phase_c()


for o in m.optimizers:
    o.sync_lr()
g0 = capture(base_step)
g1 = capture(step_overlap)
for _ in range(2):
    print('graph baseline       %.2f ms' % timeit(g0.replay, 20))
    print('graph real-overlap   %.2f ms' % timeit(g1.replay, 20))
