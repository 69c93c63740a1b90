"""Where process_volume's wall time goes (host numpy vs device): python tools/prof_volume.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hvgan
from hvgan import synth, infer
from hvgan.models.inpaint_networks import Generator

os.environ.setdefault('HV_PRECISION', 'fp16')
net = Generator({'input_dim': 1, 'ngf': 16}, True).cuda().eval()
ct, label, cam = synth.make_volume(nz=64, size=256, seed=2)
cam255 = cam * 255
dev = torch.device('cuda:0')
for _ in range(3):
    infer.process_volume(net, ct, label, cam255, 20, dev)
torch.cuda.synchronize()
T = {}
def tic(name, f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t; return r
n = 5
for _ in range(n):
    zh = tic('z-extent', lambda: np.flatnonzero((label == 20).any(axis=(0, 1))))
    slab = lambda vol: np.ascontiguousarray(vol[:, :, 6:58], dtype=np.float32)
    l = tic('slab label', lambda: slab(label)); c = tic('slab ct', lambda: slab(ct)); m = tic('slab cam', lambda: slab(cam255))
    st = tic('stack', lambda: np.stack([l, c, m]))
    d = tic('h2d', lambda: torch.from_numpy(st).to(dev))
    tic('counts', lambda: ((l == 19).sum(axis=(0, 1)) > 200, (l == 21).sum(axis=(0, 1)) > 200))
    o = tic('zeros', lambda: (np.zeros(ct.shape), np.zeros(ct.shape)))
    r = torch.zeros(2, 256 * 256, 52, device=dev)
    rh = tic('d2h', lambda: r.cpu().numpy().reshape(2, 256, 256, 52))
    def asg():
        o[0][:, :, 6:58], o[1][:, :, 6:58] = rh[0], rh[1]
    tic('assign', asg)
    tic('whole', lambda: infer.process_volume(net, ct, label, cam255, 20, dev))
for k, v in T.items():
    print('%-12s %7.2f ms' % (k, v / n * 1e3))
