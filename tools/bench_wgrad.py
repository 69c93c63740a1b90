"""Micro-benchmark of one weight-gradient shape through the C ABI (for rocprofv3 --pmc runs and tile tuning).

    python tools/bench_wgrad.py B H W Cin Cout k stride pad [iters=20] [precision=fp16]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hvgan
from hvgan import ops


def main():
    a = sys.argv[1:]
    B, H, W, Cin, Cout, k, s, p = (int(v) for v in a[:8])
    iters = int(a[8]) if len(a) > 8 else 20
    prec = a[9] if len(a) > 9 else 'fp16'
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    Ho, Wo = ops.conv_out_size(H, k, s, p, 1), ops.conv_out_size(W, k, s, p, 1)
    x = ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).to(ops.storage_dtype(prec)))
    gy = ops.Act(torch.randn(B, Ho, Wo, Cout, generator=g).to(dev).to(ops.storage_dtype(prec)))
    dw = torch.empty(Cout, k * k, Cin, device=dev)
    for _ in range(3):
        ops.conv2d_wgrad(x, gy, dw, k, s, p, 1, precision=prec)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.conv2d_wgrad(x, gy, dw, k, s, p, 1, precision=prec)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    print('W B%d %dx%d Cin%d->Cout%d k%d s%d %s: %.1f us  %.1f TF' % (B, H, W, Cin, Cout, k, s, prec, us, fl / us / 1e6))


if __name__ == '__main__':
    main()
