"""The same short training run in the fp16 storage mode and in the fp32 parity mode (same seed, same batches): loss trajectories side by side.
python tools/fp16_vs_fp32_run.py [steps] [batch] [size]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S = int(sys.argv[3]) if len(sys.argv) > 3 else 128
runs = {}
for prec in ('fp32', 'fp16'):
    os.environ['HV_PRECISION'] = prec
    torch.manual_seed(0)
    opt = bench.make_opt(prec)
    m = Pix2PixModel(opt); m.setup(opt)
    hist = []
    for step in range(N):
        m.set_input(synth.make_batch(B, S, seed=10_000 + step % 8))
        m.optimize_parameters()
        if step % 5 == 4:
            hist.append(m.get_current_losses())
    runs[prec] = hist
    print(prec, 'skipped steps', m.overflow_steps())
    del m
for i, (a, b) in enumerate(zip(runs['fp32'], runs['fp16'])):
    print('step %3d ' % (5 * i + 5) + '  '.join('%s %.3f/%.3f' % (k, a[k], b[k]) for k in a))
