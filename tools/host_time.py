"""Host-side (Python + launch) time per train step vs. device time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ.setdefault('HV_PRECISION', 'fp16')
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
torch.manual_seed(1234)
opt = bench.make_opt('fp16')
model = Pix2PixModel(opt); model.setup(opt)
model.set_input(synth.make_batch(16, 256, seed=1234))
for _ in range(3):
    model.optimize_parameters()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    model.optimize_parameters()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('host issue time per step %.2f ms; wall per step %.2f ms' % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
