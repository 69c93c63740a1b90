"""The contextual-attention block alone (forward + backward, fp16 mode) for rocprofv3 --kernel-trace --stats: per-kernel durations without the rest of the step.

    python tools/bench_attention.py [B=16] [H=64] [iters=20]        (HV_CA_GRAM=0: the patch-table route)
"""
import os
import sys

os.environ.setdefault('HV_PRECISION', 'fp16')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hvgan  # noqa: F401
from hvgan import engine, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
H = int(sys.argv[2]) if len(sys.argv) > 2 else 64
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device('cuda:0')
C = 64
f = ops.Act(torch.randn(B, H, H, C, device=dev).half())
mask = torch.zeros(B, 1, 4 * H, 4 * H, device=dev)
mask[:, :, 4 * H // 3:4 * H // 3 + 40, :] = 1
dout = ops.Act((torch.randn(B, H, H, C, device=dev) * 0.05).half())
plan = engine.AttentionPlan(B, H, H, C, dev, (4 * H, 4 * H))
out = ops.Act(torch.zeros(B, H, H, C, dtype=torch.float16, device=dev))
df = ops.Act(torch.zeros(B, H, H, C, dtype=torch.float16, device=dev))
for _ in range(3):
    plan.forward(f, mask, out, 'fp16')
    plan.backward(dout, df, False, 'fp16')
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
tf = tb = 0.0
for _ in range(iters):
    e0.record(); plan.forward(f, mask, out, 'fp16'); e1.record(); plan.backward(dout, df, False, 'fp16'); e2.record()
    torch.cuda.synchronize()
    tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
print('attention B%d %dx%d gram=%s: forward %.1f us  backward %.1f us (eager launches, events)' % (B, H, H, getattr(plan, 'gram', None), tf / iters * 1e3, tb / iters * 1e3))
