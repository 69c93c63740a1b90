import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import torch, torch.nn.functional as F
import hvgan
from hvgan import ops
from hvtest import to_act, dev
def run(B,H,W,Cin,Cout,k,p):
    g0 = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, H, W, generator=g0)
    w = torch.randn(Cout, Cin, k, k, generator=g0).requires_grad_(True)
    y = F.conv2d(x, w, None, stride=1, padding=p)
    g = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
    y.backward(g)
    CoutP = (Cout + 3)//4*4
    xa = to_act(x, Cin); xa = ops.Act(xa.t, Cin, 0)
    ga = to_act(g, CoutP); ga = ops.Act(ga.t, CoutP, 0)
    dw = torch.empty(CoutP, k*k, Cin, device=dev())
    ops.conv2d_wgrad(xa, ga, dw, k, 1, p, 1, precision='fp16')
    torch.cuda.synchronize()
    got = dw[:Cout].cpu().reshape(Cout, k, k, Cin).permute(0, 3, 1, 2)
    err = (got - w.grad).abs()
    print((B,H,W,Cin,Cout,k), 'max err', err.max().item(), 'scale', w.grad.abs().max().item())
    bad = (err > 0.05 * w.grad.abs().max()).nonzero()
    print(' bad count', len(bad), 'taps (r,q) of bad:', sorted(set((int(b[2]), int(b[3])) for b in bad))[:16], 'ci range', (int(bad[:,1].min()), int(bad[:,1].max())) if len(bad) else None)
run(2,16,16,256,1,4,1)
run(2,16,16,32,1,4,1)
run(2,16,16,16,1,4,1)
run(2,32,32,32,4,4,1)
run(2,16,16,32,1,3,1)
B,H,W,Cin,Cout,k,p = 2,16,16,32,1,4,1
g0 = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, H, W, generator=g0)
w = torch.randn(Cout, Cin, k, k, generator=g0).requires_grad_(True)
y = F.conv2d(x, w, None, stride=1, padding=p)
g = torch.randn(y.shape, generator=torch.Generator().manual_seed(2))
y.backward(g)
xa = to_act(x, Cin); xa = ops.Act(xa.t, Cin, 0)
ga = to_act(g, 4); ga = ops.Act(ga.t, 4, 0)
dw = torch.empty(4, k*k, Cin, device=dev())
ops.conv2d_wgrad(xa, ga, dw, k, 1, p, 1, precision='fp16')
got = dw[:1].cpu().reshape(1, k, k, Cin).permute(0, 3, 1, 2)
print('ref  r=0:', w.grad[0, :4, 0, :].tolist())
print('got  r=0:', got[0, :4, 0, :].tolist())
print('ref ci=20 r=1:', w.grad[0, 20, 1, :].tolist()); print('got ci=20 r=1:', got[0, 20, 1, :].tolist())
