"""Train steps fed by the device batch assembler: every step draws 16 new slices from resident synthetic volumes,
assembles the batch on the GPU and runs the full G + 3xD step.   python tools/train_device_loader.py [steps=30] [volumes=32]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
os.environ['HV_PRECISION'] = 'fp16'
import hvgan
from hvgan import synth
from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler
from hvgan.models.pix2pix_model import Pix2PixModel


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    nvol = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    raw = [synth.make_spine_volume(s, H=256, W=256, Z=48, pitch=48) for s in range(4)]
    vols = [VertebraVolume(*raw[i % 4], 11 + i % 3, ['10', '14'], path='v%d' % i) for i in range(nvol)]
    asm = DeviceBatchAssembler(vols, 'cuda:0')
    torch.manual_seed(0)
    np.random.seed(0)
    opt = bench.make_opt('fp16')
    model = Pix2PixModel(opt)
    model.setup(opt)
    order = np.random.permutation(nvol)
    host = [0.0, 0.0, 0.0]

    def step(i):
        idx = [int(order[(16 * i + j) % nvol]) for j in range(16)]
        t0 = time.perf_counter()
        b = asm.batch(idx)
        t1 = time.perf_counter()
        model.set_input(b)
        t2 = time.perf_counter()
        model.optimize_parameters()
        t3 = time.perf_counter()
        host[0] += t1 - t0; host[1] += t2 - t1; host[2] += t3 - t2
    for i in range(6):
        step(i)
    torch.cuda.synchronize()
    host[:] = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for i in range(steps):
        step(6 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('host time per step: assemble %.2f ms, set_input %.2f ms, optimize_parameters (launch) %.2f ms' % tuple(h / steps * 1e3 for h in host))
    print('train step incl. device batch assembly (16 fresh slices per step): %.2f ms/step = %.0f slices/s; losses %s'
          % (dt * 1e3, 16 / dt, {k: round(v, 3) for k, v in list(model.get_current_losses().items())[:4]}))


if __name__ == '__main__':
    main()
