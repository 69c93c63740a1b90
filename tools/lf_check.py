"""conv_lf_kernel against conv_halo2_kernel (same inputs; the LDS-filter kernel is taken when the fragment-ordered table is passed) and timing of both.

    python tools/lf_check.py B H W Cin Cout [dil] [stride]        (stride 2: forward only)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import hvgan
from hvgan import ops, lib


def main():
    a = [int(v) for v in sys.argv[1:]]
    B, H, W, Cin, Cout = a[:5]
    dil = a[5] if len(a) > 5 else 1
    st = a[6] if len(a) > 6 else 1
    Ho, Wo = (H + st - 1) // st, (W + st - 1) // st
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(0)
    x = ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).half())
    w = (torch.randn(Cout, 9, Cin, generator=g) / (Cin * 9) ** 0.5).to(dev)
    wh = w.half()
    wt = ops.tile_weights(wh, Cout, 9, Cin)
    bias = torch.randn(Cout, generator=g).to(dev)
    m = ops.Act(torch.randn(B, H, W, Cout, generator=g).to(dev).half())
    forms = ((False, dict(act='elu', bias=bias)), (True, dict(mul=(m, 'elu'))), (True, dict(mul=(m, 'elu'), accumulate=1)))
    for tr, kw in forms[:1] if st != 1 else forms:
        y0 = ops.Act(torch.full((B, Ho, Wo, Cout), 0.5, device=dev, dtype=torch.float16))
        y1 = ops.Act(torch.full((B, Ho, Wo, Cout), 0.5, device=dev, dtype=torch.float16))
        ops.conv2d(x, w, y0, 3, st, dil, dil, transposed=tr, precision='fp16', w_h=wh, **kw)
        p0 = lib.get().size('hv_last_kernel_path')
        ops.conv2d(x, w, y1, 3, st, dil, dil, transposed=tr, precision='fp16', w_h=wh, w_t=wt, **kw)
        p1 = lib.get().size('hv_last_kernel_path')
        torch.cuda.synchronize()
        d = (y0.t.float() - y1.t.float()).abs().max().item()
        print('transposed=%d %s paths %d %d  max|diff| %.3e  equal %s  |y| %.3f' % (tr, sorted(kw), p0, p1, d, torch.equal(y0.t, y1.t), y0.t.float().abs().mean().item()))
    for name, wtt in (('halo2', None), ('lf', wt)):
        y = ops.Act.empty(B, Ho, Wo, Cout, dev, dtype=torch.float16)
        run = lambda: ops.conv2d(x, w, y, 3, st, dil, dil, precision='fp16', w_h=wh, w_t=wtt, act='elu', bias=bias)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                run()
        gr.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gr.replay(); e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print('%s: %.1f us  %.0f TF' % (name, us, 2.0 * B * Ho * Wo * Cout * 9 * Cin / us / 1e6))


if __name__ == '__main__':
    main()
