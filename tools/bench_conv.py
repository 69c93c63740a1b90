"""Micro-benchmark of one convolution shape through the C ABI (for rocprofv3 --pmc runs and tile tuning).

    python tools/bench_conv.py B H W Cin Cout k stride pad [transposed=0] [iters=20] [precision=fp16]
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
    tr = int(a[8]) if len(a) > 8 else 0
    iters = int(a[9]) if len(a) > 9 else 20
    prec = a[10] if len(a) > 10 else 'fp16'
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cpu').manual_seed(0)
    if tr:
        Ho, Wo = (H - 1) * s - 2 * p + k, (W - 1) * s - 2 * p + k
        if s == 2 and k in (3, 4):      # data gradient of a pad-1 stride-2 conv on an even-sized map
            Ho, Wo = H * 2, W * 2
    else:
        dil_ = int(os.environ.get('DIL', '1'))
        Ho, Wo = ops.conv_out_size(H, k, s, p, dil_), ops.conv_out_size(W, k, s, p, dil_)
    rot = int(os.environ.get('ROTATE', '1'))       # ROTATE=n: n input/output buffer pairs visited in turn (n * size > L2 + MALL: cold reads, as inside a step)
    xs = [ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).to(ops.storage_dtype(prec))) for _ in range(rot)]
    x = xs[0]
    w = (torch.randn(Cout, k * k, Cin, generator=g) / (Cin * k * k) ** 0.5).to(dev)
    wh = w.half()
    wt = ops.tile_weights(wh, Cout, k * k, Cin) if os.environ.get('HV_W_TILED', '1') != '0' else None
    ys = [ops.Act.empty(B, Ho, Wo, Cout, dev, dtype=torch.float32 if os.environ.get('Y32') == '1' else ops.storage_dtype(prec)) for _ in range(rot)]      # Y32=1: fp32 output (the 1-channel heads)
    y = ys[0]
    dil = int(os.environ.get('DIL', '1'))          # DIL=d: dilation (pass pad = d for a same-size 3x3 layer)
    act = os.environ.get('ACT', 'none')            # epilogue as inside the step: ACT=elu BIAS=1 (forward), MUL=elu ACC=1 (data gradient with act' factor)
    bias = torch.randn(Cout, generator=g).to(dev) if os.environ.get('BIAS') == '1' else None
    mul = (ops.Act(torch.randn(B, Ho, Wo, Cout, generator=g).to(dev).to(ops.storage_dtype(prec))), os.environ['MUL']) if os.environ.get('MUL') else None
    acc = int(os.environ.get('ACC', '0'))
    run = lambda i: ops.conv2d(xs[i % rot], w, ys[i % rot], k, s, p, dil, transposed=bool(tr), precision=prec, w_h=wh, w_t=wt, act=act, bias=bias,
                               mul=mul, accumulate=acc)
    for i in range(3):
        run(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get('HV_BENCH_GRAPH', '1') != '0':      # replayed from a hipGraph: the host's launch rate is out of the measurement
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for i in range(iters):
                run(i)
        gr.replay()
        torch.cuda.synchronize()
        e0.record()
        gr.replay()
        e1.record()
    else:
        e0.record()
        for i in range(iters):
            run(i)
        e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    taps = k * k if not tr else max(1, k * k // (s * s))
    fl = 2.0 * B * Ho * Wo * Cout * taps * Cin
    chk = ''
    if os.environ.get('CHECK') == '1':      # digest of the first output buffer: A/B runs of two kernel variants must print the same
        run(0)
        torch.cuda.synchronize()
        t = ys[0].t.float()
        chk = '  sum %.6e  abs %.6e' % (t.double().sum().item(), t.double().abs().sum().item())
    print('%s B%d %dx%d Cin%d->Cout%d k%d s%d %s: %.1f us  %.1f TF%s' % ('T' if tr else 'F', B, H, W, Cin, Cout, k, s, prec, us, fl / us / 1e6, chk))


if __name__ == '__main__':
    main()
