"""Batch assembly: device-resident volumes + hv_assemble_batch vs the CPU restatement of AlignedDataset.__getitem__
(file I/O excluded on both sides).   python tools/bench_assemble.py [batch=16] [iters=50]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import hvgan
from hvgan import synth
from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler
from oracle import restate as R


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    raw = [synth.make_spine_volume(s, H=256, W=256, Z=64, pitch=48) for s in range(4)]
    t0 = time.time()
    vols = [VertebraVolume(*raw[i % 4], 12, ['11', '13'], path='v%d' % i) for i in range(B)]
    t_prep = (time.time() - t0) / B
    asm = DeviceBatchAssembler(vols, 'cuda:0')
    np.random.seed(0)
    for _ in range(5):
        asm.batch(list(range(B)))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.time()
    e0.record()
    for _ in range(iters):
        b = asm.batch(list(range(B)))
    e1.record()
    torch.cuda.synchronize()
    wall = (time.time() - t0) / iters * 1e3
    dev = e0.elapsed_time(e1) / iters
    np.random.seed(0)
    t0 = time.time()
    n = 4
    for i in range(n):
        ct, label, cam = raw[i % 4]
        R.dataset_item(ct.astype(np.float64), label.astype(np.float64), cam.astype(np.float64) * 255, 12, ['11', '13'])
    cpu = (time.time() - t0) / n * 1e3
    print('batch %d of 256x256 slices: device path %.3f ms wall per batch (GPU stream busy %.3f ms; one-off host quantisation %.1f ms per volume)'
          % (B, wall, dev, t_prep * 1e3))
    print('CPU restatement of __getitem__ (no file I/O): %.1f ms per item = %.1f ms per batch on one core -> %.0fx' % (cpu, cpu * B, cpu * B / wall))


if __name__ == '__main__':
    main()
