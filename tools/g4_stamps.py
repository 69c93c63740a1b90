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
