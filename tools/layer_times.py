"""Per-layer timing of one train step (HIP events around every conv / wgrad launch); prints the heaviest kernel classes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ['HV_PRECISION'] = sys.argv[1] if len(sys.argv) > 1 else 'fp16'
import hvgan
from hvgan import synth, profiler, engine
engine.SERIAL = True      # one stream: HIP events then time the kernel, not the wait for free CUs
from hvgan.models.pix2pix_model import Pix2PixModel
torch.manual_seed(1234)
opt = bench.make_opt(os.environ['HV_PRECISION'])
model = Pix2PixModel(opt); model.setup(opt)
model.use_graph = False
model.set_input(synth.make_batch(16, 256, seed=1234))
for _ in range(2):
    model.optimize_parameters()
torch.cuda.synchronize()
prof = profiler.KernelTimer(); prof.calibrate(); prof.enable()
model.optimize_parameters()
agg = prof.summary(); prof.disable()
tot = sum(v[0] for v in agg.values())
print('conv+wgrad total ms', round(tot, 3))
for key, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get("ROWS", "28"))]:
    print('%7.3f ms  n=%2d  avg %7.1f us  %6.1f TF  %s' % (ms, n, ms / n * 1e3, fl / (ms / n * 1e-3) / 1e12, profiler.describe(key, prof.paths.get(key))))
kinds = {}
for key, (ms, n, fl) in agg.items():
    k = key[0] + ('_T' if key[-1] else '')
    kinds[k] = kinds.get(k, 0.0) + ms
print('by kind', {k: round(v, 2) for k, v in kinds.items()})
print('--- all wgrad')
for key, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if key[0] == 'wgrad':
        print('%7.3f ms  n=%2d  avg %7.1f us  %6.1f TF  %s' % (ms, n, ms / n * 1e3, fl / (ms / n * 1e-3) / 1e12, profiler.describe(key, prof.paths.get(key))))
print('--- all conv')
for key, (ms, n, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    if key[0] == 'conv':
        print('%7.3f ms  n=%2d  avg %7.1f us  %6.1f TF  %s' % (ms, n, ms / n * 1e3, fl / (ms / n * 1e-3) / 1e12, profiler.describe(key, prof.paths.get(key))))
