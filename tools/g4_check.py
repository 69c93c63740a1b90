"""conv_g4_kernel (4x4 stride-2 forward / data gradient as pipelined implicit GEMM) against the halo kernels and torch CPU fp32.
    python tools/g4_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import hvgan
from hvgan import ops, lib

dev = torch.device('cuda:0')


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


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


for (B, H, W, Cin, Cout, ref) in [(2, 20, 24, 32, 128, 1), (3, 36, 28, 64, 256, 1), (16, 128, 128, 64, 128, 0), (16, 64, 64, 128, 256, 0), (16, 32, 32, 256, 512, 0)]:
    g = torch.Generator().manual_seed(B + H + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / (Cin * 16) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    xa = ops.Act(nhwc(x).to(dev).half())
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 16, Cin).contiguous().to(dev)
    wh = wf.half(); wt = ops.tile_weights(wh, Cout, 16, Cin)
    Ho, Wo = H // 2, W // 2
    y0, y1 = (ops.Act(torch.zeros(B, Ho, Wo, Cout, device=dev, dtype=torch.float16)) for _ in range(2))
    ops.conv2d(xa, wf, y0, 4, 2, 1, 1, precision='fp16', w_h=wh, act='lrelu', bias=bias.to(dev)); p0 = lib.get().size('hv_last_kernel_path')
    ops.conv2d(xa, wf, y1, 4, 2, 1, 1, precision='fp16', w_h=wh, w_t=wt, act='lrelu', bias=bias.to(dev)); p1 = lib.get().size('hv_last_kernel_path')
    torch.cuda.synchronize()
    d = (y0.t.float() - y1.t.float()).abs().max().item()
    msg = 'F %2d %3dx%3d %3d->%3d paths %d/%d  max|halo-g4| %.2e (|y| %.2f)' % (B, H, W, Cin, Cout, p0, p1, d, y0.t.float().abs().mean().item())
    if ref:
        r = F.leaky_relu(F.conv2d(x.half().float(), w.half().float(), bias, stride=2, padding=1), 0.2)
        msg += '  vs torch %.2e' % (nhwc(r) - y1.t.float().cpu()).abs().max().item()
    t0 = timeit(lambda: ops.conv2d(xa, wf, y0, 4, 2, 1, 1, precision='fp16', w_h=wh))
    t1 = timeit(lambda: ops.conv2d(xa, wf, y1, 4, 2, 1, 1, precision='fp16', w_h=wh, w_t=wt))
    fl = 2.0 * B * Ho * Wo * Cout * 16 * Cin
    print(msg + '   halo %.1f us  g4 %.1f us (%.0f TF)' % (t0, t1, fl / t1 / 1e6), flush=True)

# data gradient: g [B, Hg, Wg, Cg] -> dx [B, 2Hg, 2Wg, Co] with the transposed table [Co][16][Cg]
for (B, Hg, Wg, Cg, Co, ref) in [(2, 10, 12, 128, 64, 1), (3, 18, 14, 256, 128, 1), (16, 64, 64, 128, 64, 0), (16, 32, 32, 256, 128, 0), (16, 16, 16, 512, 256, 0)]:
    g = torch.Generator().manual_seed(B + Hg + Cg)
    gy = torch.randn(B, Cg, Hg, Wg, generator=g)
    w = torch.randn(Cg, Co, 4, 4, generator=g) / (Cg * 4) ** 0.5            # forward conv Co -> Cg
    m = torch.randn(B, Co, 2 * Hg, 2 * Wg, generator=g)
    ga = ops.Act(nhwc(gy).to(dev).half())
    ma = ops.Act(nhwc(m).to(dev).half())
    wb = w.permute(1, 2, 3, 0).reshape(Co, 16, Cg).contiguous().to(dev)    # [ci][taps][co]
    wh = wb.half(); wt = ops.tile_weights(wh, Co, 16, Cg)
    res = []
    for kw in (dict(), dict(mul=(ma, 'lrelu')), dict(mul=(ma, 'lrelu'), accumulate=1)):
        y0, y1 = (ops.Act(torch.full((B, 2 * Hg, 2 * Wg, Co), 0.25, device=dev, dtype=torch.float16)) for _ in range(2))
        ops.conv2d(ga, wb, y0, 4, 2, 1, 1, transposed=True, precision='fp16', w_h=wh, **kw); p0 = lib.get().size('hv_last_kernel_path')
        ops.conv2d(ga, wb, y1, 4, 2, 1, 1, transposed=True, precision='fp16', w_h=wh, w_t=wt, **kw); p1 = lib.get().size('hv_last_kernel_path')
        torch.cuda.synchronize()
        res.append('%d/%d %.2e' % (p0, p1, (y0.t.float() - y1.t.float()).abs().max().item()))
    msg = 'T %2d %3dx%3d %3d->%3d  %s (|y| %.2f)' % (B, Hg, Wg, Cg, Co, '  '.join(res), y0.t.float().abs().mean().item())
    if ref:
        y2 = ops.Act(torch.zeros(B, 2 * Hg, 2 * Wg, Co, device=dev, dtype=torch.float16))
        ops.conv2d(ga, wb, y2, 4, 2, 1, 1, transposed=True, precision='fp16', w_h=wh, w_t=wt)
        torch.cuda.synchronize()
        r = F.conv_transpose2d(gy.half().float(), w.half().float(), None, stride=2, padding=1)
        msg += '  vs torch %.2e' % (nhwc(r) - y2.t.float().cpu()).abs().max().item()
    y0 = ops.Act.empty(B, 2 * Hg, 2 * Wg, Co, dev, dtype=torch.float16)
    t0 = timeit(lambda: ops.conv2d(ga, wb, y0, 4, 2, 1, 1, transposed=True, precision='fp16', w_h=wh, mul=(ma, 'lrelu')))
    t1 = timeit(lambda: ops.conv2d(ga, wb, y0, 4, 2, 1, 1, transposed=True, precision='fp16', w_h=wh, w_t=wt, mul=(ma, 'lrelu')))
    fl = 2.0 * B * 4 * Hg * Wg * Co * 4 * Cg
    print(msg + '   halo %.1f us  g4 %.1f us (%.0f TF)' % (t0, t1, fl / t1 / 1e6), flush=True)

# ---- stride-1 4x4 (pad 1): forward Cin -> Cout, output (H-1) x (W-1); data gradient back to H x W
for (B, H, W, Cin, Cout, ref) in [(2, 11, 13, 128, 128, 1), (3, 20, 17, 32, 256, 1), (16, 32, 32, 256, 512, 0)]:
    g = torch.Generator().manual_seed(B + H + Cin + 1)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 4, 4, generator=g) / (Cin * 16) ** 0.5
    bias = torch.randn(Cout, generator=g) * 0.1
    xa = ops.Act(nhwc(x).to(dev).half())
    wf = w.permute(0, 2, 3, 1).reshape(Cout, 16, Cin).contiguous().to(dev)
    wh = wf.half(); wt = ops.tile_weights(wh, Cout, 16, Cin)
    Ho, Wo = H - 1, W - 1
    y0, y1 = (ops.Act(torch.zeros(B, Ho, Wo, Cout, device=dev, dtype=torch.float16)) for _ in range(2))
    ops.conv2d(xa, wf, y0, 4, 1, 1, 1, precision='fp16', w_h=wh, act='lrelu', bias=bias.to(dev)); p0 = lib.get().size('hv_last_kernel_path')
    ops.conv2d(xa, wf, y1, 4, 1, 1, 1, precision='fp16', w_h=wh, w_t=wt, act='lrelu', bias=bias.to(dev)); p1 = lib.get().size('hv_last_kernel_path')
    torch.cuda.synchronize()
    msg = 'F1 %2d %3dx%3d %3d->%3d paths %d/%d  max|halo-g4| %.2e (|y| %.2f)' % (B, H, W, Cin, Cout, p0, p1, (y0.t.float() - y1.t.float()).abs().max().item(), y0.t.float().abs().mean().item())
    if ref:
        r = F.leaky_relu(F.conv2d(x.half().float(), w.half().float(), bias, stride=1, padding=1), 0.2)
        msg += '  vs torch %.2e' % (nhwc(r) - y1.t.float().cpu()).abs().max().item()
    t0 = timeit(lambda: ops.conv2d(xa, wf, y0, 4, 1, 1, 1, precision='fp16', w_h=wh))
    t1 = timeit(lambda: ops.conv2d(xa, wf, y1, 4, 1, 1, 1, precision='fp16', w_h=wh, w_t=wt))
    print(msg + '   halo %.1f us  g4 %.1f us (%.0f TF)' % (t0, t1, 2.0 * B * Ho * Wo * Cout * 16 * Cin / t1 / 1e6), flush=True)
    # data gradient of the same layer: g [B, Ho, Wo, Cout] -> dx [B, H, W, Cin]
    gy = torch.randn(B, Cout, Ho, Wo, generator=g)
    m = torch.randn(B, Cin, H, W, generator=g)
    ga = ops.Act(nhwc(gy).to(dev).half()); ma = ops.Act(nhwc(m).to(dev).half())
    wb = w.permute(1, 2, 3, 0).reshape(Cin, 16, Cout).contiguous().to(dev)
    wbh = wb.half(); wbt = ops.tile_weights(wbh, Cin, 16, Cout)
    if Cin % 128:
        continue
    res = []
    for kw in (dict(), dict(mul=(ma, 'lrelu'), accumulate=1)):
        d0, d1 = (ops.Act(torch.full((B, H, W, Cin), 0.25, device=dev, dtype=torch.float16)) for _ in range(2))
        ops.conv2d(ga, wb, d0, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wbh, **kw); p0 = lib.get().size('hv_last_kernel_path')
        ops.conv2d(ga, wb, d1, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wbh, w_t=wbt, **kw); p1 = lib.get().size('hv_last_kernel_path')
        torch.cuda.synchronize()
        res.append('%d/%d %.2e' % (p0, p1, (d0.t.float() - d1.t.float()).abs().max().item()))
    msg = 'T1 %2d %3dx%3d %3d->%3d  %s' % (B, Ho, Wo, Cout, Cin, '  '.join(res))
    if ref:
        d2 = ops.Act(torch.zeros(B, H, W, Cin, device=dev, dtype=torch.float16))
        ops.conv2d(ga, wb, d2, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wbh, w_t=wbt)
        torch.cuda.synchronize()
        r = F.conv_transpose2d(gy.half().float(), w.half().float(), None, stride=1, padding=1)
        msg += '  vs torch %.2e' % (nhwc(r) - d2.t.float().cpu()).abs().max().item()
    t0 = timeit(lambda: ops.conv2d(ga, wb, d0, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wbh, mul=(ma, 'lrelu')))
    t1 = timeit(lambda: ops.conv2d(ga, wb, d1, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wbh, w_t=wbt, mul=(ma, 'lrelu')))
    print(msg + '   halo %.1f us  g4 %.1f us (%.0f TF)' % (t0, t1, 2.0 * B * H * W * Cin * 16 * Cout / t1 / 1e6), flush=True)
