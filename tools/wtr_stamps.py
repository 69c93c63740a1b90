"""Phase sums of one wgrad_tr_kernel workgroup (diagnostic build: HV_EXTRA_FLAGS=-DWT_STAMPS; the kernel prints them).  python tools/wtr_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan
from hvgan import ops
dev = torch.device('cuda:0')
for (B, H, W, Cin, Cout, k, s, p) in [(16, 31, 31, 256, 512, 4, 1, 1), (16, 128, 128, 64, 128, 4, 2, 1), (16, 64, 64, 128, 256, 4, 2, 1), (16, 64, 64, 64, 64, 3, 1, 1)]:
    g = torch.Generator().manual_seed(0)
    Ho = (H + 2 * p - k) // s + 1
    x = ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).half())
    gy = ops.Act(torch.randn(B, Ho, Ho, Cout, generator=g).to(dev).half())
    dw = torch.empty(Cout, k * k, Cin, device=dev)
    for _ in range(3):
        ops.conv2d_wgrad(x, gy, dw, k, s, p, 1, precision='fp16')
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d_wgrad(x, gy, dw, k, s, p, 1, precision='fp16')
    e1.record(); torch.cuda.synchronize()
    print('B%d %dx%d %d->%d k%d s%d: %.1f us per call (kernel + reduce)' % (B, H, W, Cin, Cout, k, s, e0.elapsed_time(e1) / 5 * 1e3), flush=True)
