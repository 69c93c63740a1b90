"""Magnitude census of the activation-gradient buffers of one bs=16 train step in the fp16 mode (fp16 storage): per network the
largest |value| and the share of non-zero elements that are subnormal in fp16 (< 6.1e-5) -- what a gradient (loss) scale has to fix,
and how much head room below 65504 it leaves.  HV_GRAD_SCALE=1 shows the unscaled picture."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HV_PRECISION', 'fp16')
os.environ['HV_GRAPH'] = '0'
import torch
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from test_step_gpu import make_opt

torch.manual_seed(1234)
model = Pix2PixModel(make_opt())
for step in range(int(os.environ.get('STEPS', '3'))):
    model.set_input(synth.make_batch(16, 256, seed=1234 + step))
    model.optimize_parameters()
    torch.cuda.synchronize()
    rows = []
    for name in ('G', 'D_1', 'D_2', 'D_3'):
        net = getattr(model, 'net' + name)
        mx, sub, nz, inf = 0.0, 0, 0, 0
        for P in net._plans.values():
            for t in P.book.twins.values():
                if t.dtype != torch.float16:
                    continue
                a = t.float().abs()
                mx = max(mx, float(a[torch.isfinite(a)].max()) if a.numel() else 0.0)
                inf += int((~torch.isfinite(a)).sum())
                n = a > 0
                nz += int(n.sum()); sub += int((n & (a < 6.1e-5)).sum())
        rows.append('%s: max |g| %.3e, subnormal share %.1f %%, non-finite %d' % (name, mx, 100.0 * sub / max(nz, 1), inf))
    print('step', step, ' | '.join(rows), flush=True)
print({k: round(v, 4) for k, v in model.get_current_losses().items()})
