"""conv_lf_kernel against conv_halo2_kernel over the generator's 3x3 stride-1 shapes: equality of the outputs and time of both.
    python tools/lf_sweep.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan
from hvgan import ops, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
#        H    Cin Cout dil shift
SHAPES = [(64, 64, 64, 1, 0), (64, 64, 64, 2, 0), (64, 64, 64, 4, 0), (64, 32, 64, 1, 0), (64, 64, 128, 1, 0), (64, 64, 32, 1, 0),
          (128, 32, 32, 1, 0), (128, 64, 32, 1, 0), (128, 64, 32, 1, 1), (128, 16, 32, 1, 0), (128, 32, 64, 1, 0), (128, 32, 16, 1, 0),
          (256, 32, 16, 1, 0), (256, 32, 16, 1, 1), (256, 16, 8, 1, 0), (256, 16, 32, 1, 0), (256, 16, 16, 1, 0)]


def timeit(run):
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(10):
            run()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


for H, Cin, Cout, dil, shift in SHAPES:
    g = torch.Generator().manual_seed(H + Cin + Cout)
    Hi = H >> shift
    x = ops.Act(torch.randn(B, Hi, Hi, Cin, generator=g).to(dev).half())
    w = (torch.randn(Cout, 9, Cin, generator=g) / (Cin * 9) ** 0.5).to(dev)
    wh = w.half(); wt = ops.tile_weights(wh, Cout, 9, Cin)
    bias = torch.randn(Cout, generator=g).to(dev)
    m = ops.Act(torch.randn(B, H, H, Cout, generator=g).to(dev).half())
    res = []
    forms = [(False, dict(act='elu', bias=bias, in_shift=shift))]
    if not shift:
        forms += [(True, dict(mul=(m, 'elu'))), (True, dict(mul=(m, 'lrelu'), accumulate=1))]
    for tr, kw in forms:
        y0 = ops.Act(torch.full((B, H, H, Cout), 0.5, device=dev, dtype=torch.float16))
        y1 = ops.Act(torch.full((B, H, H, Cout), 0.5, device=dev, dtype=torch.float16))
        ops.conv2d(x, w, y0, 3, 1, dil, dil, transposed=tr, precision='fp16', w_h=wh, **kw)
        p0 = lib.get().size('hv_last_kernel_path')
        ops.conv2d(x, w, y1, 3, 1, dil, dil, transposed=tr, precision='fp16', w_h=wh, w_t=wt, **kw)
        p1 = lib.get().size('hv_last_kernel_path')
        torch.cuda.synchronize()
        res.append('%d/%d:%s' % (p0, p1, 'eq' if torch.equal(y0.t, y1.t) else 'DIFF %.2e' % (y0.t.float() - y1.t.float()).abs().max().item()))
    y = ops.Act.empty(B, H, H, Cout, dev, dtype=torch.float16)
    t0 = timeit(lambda: ops.conv2d(x, w, y, 3, 1, dil, dil, precision='fp16', w_h=wh, act='elu', bias=bias, in_shift=shift))
    t1 = timeit(lambda: ops.conv2d(x, w, y, 3, 1, dil, dil, precision='fp16', w_h=wh, w_t=wt, act='elu', bias=bias, in_shift=shift))
    print('%3d^2 %3d->%3d d%d shift%d  %-40s  halo2 %6.1f us  lf %6.1f us  (%.2fx)' % (H, Cin, Cout, dil, shift, ' '.join(res), t0, t1, t0 / t1), flush=True)
