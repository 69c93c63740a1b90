"""Deviation of the fp16-MFMA mode from the reference's golden two-step run (G5)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ['HV_PRECISION'] = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
import torch
from conftest import load_golden
from test_step_gpu import make_opt, _sparse
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
g = load_golden('g5_full_step')
torch.manual_seed(1234)
model = Pix2PixModel(make_opt())
for step in range(2):
    model.set_input(synth.make_batch(2, 256, seed=1234 + step))
    model.optimize_parameters()
    torch.cuda.synchronize()
    losses = model.get_current_losses()
    print('step', step, 'loss rel dev', {k: round(abs(losses[k] - float(v)) / max(1.0, abs(float(v))), 5) for k, v in g['losses%d' % step].items()})
    for k, ref in g['samples%d' % step].items():
        got = _sparse(getattr(model, k))
        d = (got - ref).abs()
        print('   %-22s max %.4f  frac>1e-2 %.5f' % (k, d.max().item(), (d > 1e-2).float().mean().item()))
    worst = 0
    for key, ref in g['norms%d' % step].items():
        n, k = key.split('/', 1)
        got = float(getattr(model, 'net' + n).state_dict()[k].double().norm())
        worst = max(worst, abs(got - float(ref)) / max(1.0, float(ref)))
    print('   worst param-norm rel dev', worst)
