"""Phase times of one conv_lf_kernel workgroup (diagnostic build: HV_EXTRA_FLAGS=-DLF_STAMPS python healthivert-gan_amd/csrc/build.py after touching conv_lf.hip).
    python tools/lf_stamps.py B H W Cin Cout"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan
from hvgan import ops

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
x = ops.Act(torch.randn(B, H, W, Cin, generator=g).to(dev).half())
w = (torch.randn(Cout, 9, Cin, generator=g) / (Cin * 9) ** 0.5).to(dev)
wh = w.half(); wt = ops.tile_weights(wh, Cout, 9, Cin)
bias = torch.randn(Cout, generator=g).to(dev)
y = ops.Act.empty(B, H, W, Cout, dev, dtype=torch.float16)
names = ['prologue+issue', 'loads->LDS+barrier', 'mfma loop', 'barrier+stage', 'LDS->stores']
for rep in range(4):
    for _ in range(3):
        ops.conv2d(x, w, y, 3, 1, 1, 1, precision='fp16', w_h=wh, w_t=wt, act='elu', bias=bias)
    torch.cuda.synchronize()
    st = y.t.view(-1)[:24].view(torch.int64).cpu().tolist()
    print('  '.join('%s %.2f us' % (n, (st[i + 1] - st[i]) / 100.0) for i, n in enumerate(names)), ' total %.2f' % ((st[5] - st[0]) / 100.0))
