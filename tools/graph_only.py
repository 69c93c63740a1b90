"""K captured-graph train steps and nothing else (for rocprofv3 --kernel-trace timelines of the replayed step):
    rocprofv3 --kernel-trace --output-format csv -d out -o tr -- python3 tools/graph_only.py [steps=8]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
os.environ['HV_PRECISION'] = 'fp16'
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
torch.manual_seed(1234)
opt = bench.make_opt('fp16')
model = Pix2PixModel(opt)
model.setup(opt)
model.set_input(synth.make_batch(16, 256, seed=1234))
for _ in range(model.GRAPH_WARMUP + 2):
    model.optimize_parameters()
torch.cuda.synchronize()
assert model._graphs is not None
for _ in range(steps):
    model.optimize_parameters()
torch.cuda.synchronize()
