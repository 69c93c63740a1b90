"""Which copy_channels / act_backward passes does one train step launch, from where?  python tools/glue_calls.py"""
import os, sys, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ['HV_PRECISION'] = 'fp16'
import hvgan
from hvgan import synth, ops, engine
engine.SERIAL = True
from hvgan.models.pix2pix_model import Pix2PixModel
torch.manual_seed(1234)
opt = bench.make_opt('fp16')
model = Pix2PixModel(opt); model.setup(opt)
model.use_graph = False
model.set_input(synth.make_batch(16, 256, seed=1234))
for _ in range(2):
    model.optimize_parameters()
torch.cuda.synchronize()
log = collections.Counter()
def where():
    out = []
    for fr in reversed(traceback.extract_stack()[:-2]):
        if 'tools/' in fr.filename or fr.name in ('cc', 'ab') or 'torch' in fr.filename:
            continue
        out.append('%s:%d' % (os.path.basename(fr.filename).replace('.py', ''), fr.lineno))
        if len(out) == 3:
            break
    return ' < '.join(out)
cc0, ab0 = ops.copy_channels, ops.act_backward
def cc(src, dst, mode=0, accumulate=False):
    log[('copy', where(), (src.B, src.H, src.W, src.C, src.ld, 'h' if src.f16 else 'f'), (dst.H, dst.W, dst.C, dst.ld, 'h' if dst.f16 else 'f'), mode, bool(accumulate))] += 1
    return cc0(src, dst, mode, accumulate)
def ab(dy, y, act, dbias=None, dbias_accumulate=False):
    log[('act_bwd', where(), (dy.B, dy.H, dy.W, dy.C, dy.ld, 'h' if dy.f16 else 'f'), act, dbias is not None)] += 1
    return ab0(dy, y, act, dbias, dbias_accumulate)
ops.copy_channels, ops.act_backward = cc, ab
for m in list(sys.modules.values()):
    if m is not None and getattr(m, '__name__', '').startswith(('hvgan', 'healthivert')):
        if getattr(m, 'copy_channels', None) is cc0: m.copy_channels = cc
        if getattr(m, 'act_backward', None) is ab0: m.act_backward = ab
model.optimize_parameters()
torch.cuda.synchronize()
def mb(k):
    s = k[2]
    return s[0] * s[1] * s[2] * s[3] * (2 if s[5] == 'h' else 4) / 1e6
for k, n in sorted(log.items(), key=lambda kv: -mb(kv[0]) * kv[1]):
    print('%2d x %6.1f MB  %s' % (n, mb(k), k))
