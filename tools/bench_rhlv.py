"""RHLV quantification of one 256 x 256 x 64 label-volume pair: device (hv_rhlv, volumes resident in HBM) vs the CPU oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import hvgan
from hvgan import evaluation, synth
from oracle import restate as R

fake, label = synth.make_rhlv_pair(seed=7, H=256, W=256, Z=64, empty_ends=6)
f, l = torch.from_numpy(fake).float().cuda(), torch.from_numpy(label).float().cuda()
for _ in range(3):
    evaluation._run(f, l, 20.0, 5, evaluation.INT_MIN, 0, 0.64)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
N = 50
for _ in range(N):
    evaluation._run(f, l, 20.0, 5, evaluation.INT_MIN, 0, 0.64)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / N * 1e3
t0 = time.perf_counter()
ref, _ = R.rhlv_volume(fake, label, 20)
cpu_ms = (time.perf_counter() - t0) * 1e3
got = evaluation.rhlv_volume(f, l, 20)
byt = 2 * fake.size * 4
print('hv_rhlv 256x256x64 pair: %.1f us per pair (%.1f GB/s of the %.1f MB read once), CPU oracle %.1f ms; max |d| %.2e'
      % (us, byt / us / 1e3, byt / 1e6, cpu_ms, max(abs(a - b) for a, b in zip(got, ref))))
