"""Which torch fills (zeros / zeros_like / zero_ / fill_ / full) does one steady-state train step issue, from where?  python tools/fill_calls.py"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ['HV_PRECISION'] = 'fp16'
import hvgan
from hvgan import synth, engine
from hvgan.models.pix2pix_model import Pix2PixModel
torch.manual_seed(1234)
opt = bench.make_opt('fp16')
model = Pix2PixModel(opt); model.setup(opt)
model.use_graph = False
model.set_input(synth.make_batch(16, 256, seed=1234))
for _ in range(3):
    model.optimize_parameters()
torch.cuda.synchronize()
log = collections.Counter()
def where():
    out = []
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'tools/' in fr.filename or '/torch/' in fr.filename:
            continue
        out.append('%s:%d' % (os.path.basename(fr.filename).replace('.py', ''), fr.lineno))
        if len(out) == 3:
            break
    return ' < '.join(out)
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        r = f(*a, **k)
        t = r if isinstance(r, torch.Tensor) else (a[0] if a and isinstance(a[0], torch.Tensor) else None)
        if t is not None and t.is_cuda:
            log[(name, where(), tuple(t.shape), str(t.dtype))] += 1
        return r
    setattr(mod, name, g)
for n in ('zeros', 'zeros_like', 'full', 'ones', 'full_like', 'ones_like'):
    wrap(torch, n)
for n in ('zero_', 'fill_', 'copy_', 'clone', 'float', 'half', 'mul_', 'add_', 'sum', 'mean'):      # (copies and stray element-wise ops too)
    wrap(torch.Tensor, n)
model.optimize_parameters()
torch.cuda.synchronize()
for k, v in sorted(log.items(), key=lambda kv: -kv[1]):
    print(v, k)
