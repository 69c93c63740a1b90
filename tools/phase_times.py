"""Wall time of the phases of one train step, launched eagerly (HIP events on the main stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ.setdefault('HV_PRECISION', 'fp16')
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
torch.manual_seed(1234)
opt = bench.make_opt('fp16')
m = Pix2PixModel(opt); m.setup(opt)
m.use_graph = False
m.set_input(synth.make_batch(16, 256, seed=1234))
for _ in range(3):
    m.optimize_parameters()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
acc = [0.0] * 6
N = 10
for it in range(N):
    main = torch.cuda.current_stream()
    ev[0].record()
    m.forward()
    ev[1].record()
    # phase A without the forward
    m._dxs = {}
    for k, bw in ((1, m.backward_D_1), (2, m.backward_D_2), (3, m.backward_D_3)):
        side = m._d_streams[k - 1]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            bw()
    m._join_d(main)
    ev[2].record()
    for k in (1, 2, 3):
        side = m._d_streams[k - 1]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            getattr(m, 'optimizer_D_%d' % k).step(sync_lr=False)
            m._g_step_D(k)
    m._join_d(main)
    ev[3].record()
    m.backward_G(d_done=True)
    ev[4].record()
    m.optimizer_G.step(sync_lr=False)
    ev[5].record()
    torch.cuda.synchronize()
    for i in range(5):
        acc[i] += ev[i].elapsed_time(ev[i + 1])
print('G forward+post %.2f ms | 3x D fwd/bwd fake+real (3 streams) %.2f ms | 3x D adam + D(fake) fwd/bwd for G %.2f ms | G losses+backward %.2f ms | G adam %.2f ms'
      % tuple(a / N for a in acc[:5]))
