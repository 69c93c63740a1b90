"""Eager launches vs captured graphs on the same batches: every netG tensor must come out bit-identical (debug aid for
tests/test_step_gpu.py::test_graph_recapture_when_the_batch_shape_changes).  Prints the tensors that differ."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.setdefault('HV_PRECISION', 'fp16')
import torch
import hvgan  # noqa: F401
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt


def run(use_graph, sizes, poison=False):
    from hvgan import ops
    torch.manual_seed(7)
    model = Pix2PixModel(make_opt())
    model.use_graph = use_graph
    for step, B in enumerate(sizes):
        model.set_input(synth.make_batch(B, 256, seed=900 + step))
        if poison:      # a kernel that reads scratch memory it did not write in the same call now reads NaNs (0xFF bytes)
            torch.cuda.synchronize()
            for b in list(ops.WS.buf.values()) + ops.WS.retired:
                b.fill_(255)
        model.optimize_parameters()
    torch.cuda.synchronize()
    out = {}
    for n in ('G', 'D_1', 'D_2', 'D_3'):
        for k, v in getattr(model, 'net' + n).state_dict().items():
            out[n + '.' + k] = v.detach().clone()
    return out


if os.environ.get('PREAMBLE'):      # what the test file does first: other precisions / batch sizes leave their pools and caches behind
    for prec, B in [('fp32', 2), ('fp16', 2)] * int(os.environ['PREAMBLE']):
        os.environ['HV_PRECISION'] = prec
        m = Pix2PixModel(make_opt())
        m.set_input(synth.make_batch(B, 256, seed=5))
        m.optimize_parameters()
        torch.cuda.synchronize()
        del m
    os.environ['HV_PRECISION'] = 'fp16'

sizes = tuple(int(a) for a in sys.argv[1].split(',')) if len(sys.argv) > 1 else (2, 2, 2, 2, 1, 1, 1, 2, 2, 2)
a = run(False, sizes)
for tag, other in (('eager again', run(False, sizes)), ('eager, scratch buffers poisoned before every step', run(False, sizes, True)),
                   ('graph', run(True, sizes))):
    bad = [(k, (a[k].float() - other[k].float()).abs().max().item()) for k in a if not torch.equal(a[k], other[k])]
    print('%s: %d of %d tensors differ' % (tag, len(bad), len(a)))
    for k, d in bad[:12]:
        print('   ', k, d)
