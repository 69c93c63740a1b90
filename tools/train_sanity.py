"""Short synthetic training run (graph replay, fp16 mode): losses must stay finite; prints them every 10 steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
import torch
import bench
os.environ.setdefault('HV_PRECISION', 'fp16')
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel

torch.manual_seed(0)
opt = bench.make_opt('fp16')
m = Pix2PixModel(opt); m.setup(opt)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
for step in range(N):
    m.set_input(synth.make_batch(16, 256, seed=10_000 + step % 8))
    m.optimize_parameters()
    if step % 10 == 9 or step == N - 1:
        L = m.get_current_losses()
        assert all(math.isfinite(v) for v in L.values()), L
        print(step + 1, ' '.join('%s %.3f' % kv for kv in L.items()), flush=True)
print('ok: %d steps, graphs %s' % (N, m._graphs is not None))
