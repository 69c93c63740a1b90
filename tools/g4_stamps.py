"""Phase times of one conv_g4_kernel workgroup (diagnostic build: HV_EXTRA_FLAGS=-DG4_STAMPS).  python tools/g4_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan
from hvgan import ops
dev = torch.device('cuda:0')
names = ['prologue+issue', 'first flush+barrier', 'main loop', 'stage', 'stores']
for (tr, B, H, W, Cin, Cout) in [(0, 16, 128, 128, 64, 128), (1, 16, 32, 32, 256, 128)]:
    g = torch.Generator().manual_seed(0)
    x = ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).half())
    w = (torch.randn(Cout, 16, Cin, generator=g) / (Cin * 16) ** 0.5).to(dev)
    wh = w.half(); wt = ops.tile_weights(wh, Cout, 16, Cin)
    Ho = H * 2 if tr else H // 2
    y = ops.Act.empty(B, Ho, Ho, Cout, dev, dtype=torch.float16)
    for rep in range(2):
        for _ in range(3):
            ops.conv2d(x, w, y, 4, 2, 1, 1, transposed=bool(tr), precision='fp16', w_h=wh, w_t=wt)
        torch.cuda.synchronize()
        st = y.t.view(-1)[:24].view(torch.int64).cpu().tolist()
        print(tr, Cin, Cout, '  '.join('%s %.2f us' % (n, (st[i + 1] - st[i]) / 100.0) for i, n in enumerate(names)), ' total %.2f' % ((st[5] - st[0]) / 100.0))

# the stride-1 256 <-> 512 layers (conv_g4s1_kernel): prologue to the first barrier, K loop, epilogue up to the last store's issue, store drain
names1 = ['prologue -> first data', 'K loop', 'epilogue', 'store drain']
for (tr, B, H, W, Cin, Cout) in [(0, 16, 32, 32, 256, 512), (0, 32, 32, 32, 256, 512), (1, 16, 31, 31, 512, 256)]:
    g = torch.Generator().manual_seed(0)
    x = ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).half())
    w = (torch.randn(Cout, 16, Cin, generator=g) / (Cin * 16) ** 0.5).to(dev)
    wh = w.half(); wt = ops.tile_weights(wh, Cout, 16, Cin)
    Ho = H + 1 if tr else H - 1
    y = ops.Act.empty(B, Ho, Ho, Cout, dev, dtype=torch.float16)
    bias = None if tr else torch.randn(Cout, generator=g).to(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        for _ in range(3):
            ops.conv2d(x, w, y, 4, 1, 1, 1, transposed=bool(tr), precision='fp16', w_h=wh, w_t=wt, act='none' if tr else 'lrelu', bias=bias)
        e0.record()
        ops.conv2d(x, w, y, 4, 1, 1, 1, transposed=bool(tr), precision='fp16', w_h=wh, w_t=wt, act='none' if tr else 'lrelu', bias=bias)
        e1.record()
        torch.cuda.synchronize()
        st = y.t.view(-1)[:20].view(torch.int64).cpu().tolist()
        print('s1', tr, B, Cin, Cout, '  '.join('%s %.2f us' % (n, (st[i + 1] - st[i]) / 100.0) for i, n in enumerate(names1)),
              ' in-kernel %.2f  launch by events %.1f' % ((st[4] - st[0]) / 100.0, e0.elapsed_time(e1) * 1e3))
