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
    chk = ''
    if os.environ.get('CHECK') == '1':      # against torch's weight gradient of the stored (rounded) operands, fp32
        xr, gr = x.t.float().permute(0, 3, 1, 2).contiguous(), gy.t.float().permute(0, 3, 1, 2).contiguous()
        ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, k, k), gr, stride=s, padding=p)      # [Cout][Cin][k][k]
        ref = ref.permute(0, 2, 3, 1).reshape(Cout, k * k, Cin)
        chk = '  max|d| %.3e of %.3e' % ((dw - ref).abs().max().item(), ref.abs().max().item())
    print('W B%d %dx%d Cin%d->Cout%d k%d s%d %s: %.1f us  %.1f TF%s' % (B, H, W, Cin, Cout, k, s, prec, us, fl / us / 1e6, chk))


if __name__ == '__main__':
    main()
