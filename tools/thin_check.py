"""thin_dgrad_kernel (1-channel heads' data gradient) against conv_halo2_kernel (HV_THIN_DGRAD=0 in another process is the A/B; here: vs torch CPU) + time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import hvgan
from hvgan import ops, lib
dev = torch.device('cuda:0')
for (B, H, Cout) in [(2, 40, 8), (2, 33, 12), (16, 256, 8), (16, 256, 12)]:
    g_ = torch.Generator().manual_seed(Cout)
    gy = torch.zeros(B, 4, H, H); gy[:, 0] = torch.randn(B, H, H, generator=g_)
    w = torch.randn(1, Cout, 3, 3, generator=g_) / 3.0                      # forward conv Cout -> 1
    m = torch.randn(B, Cout, H, H, generator=g_)
    ga = ops.Act(gy.permute(0, 2, 3, 1).contiguous().to(dev).half(), 4, 0)
    ma = ops.Act(m.permute(0, 2, 3, 1).contiguous().to(dev).half())
    wb = torch.zeros(Cout, 9, 4); wb[:, :, 0] = w[0].reshape(Cout, 9)       # [ci=Cout rows][taps][coP = 4]
    wb = wb.to(dev)
    y = ops.Act(torch.full((B, H, H, Cout), 0.25, device=dev, dtype=torch.float16))
    ops.conv2d(ga, wb, y, 3, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(ma, 'elu'), accumulate=1, cin=4)
    path = lib.get().size('hv_last_kernel_path')
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(gy[:, :1].half().float(), w.half().float() if False else w, None, stride=1, padding=1)
    mh = m.half().float()
    want = 0.25 + ref * torch.where(mh > 0, torch.ones_like(mh), mh + 1)
    err = (y.t.float().cpu().permute(0, 3, 1, 2) - want).abs().max().item()
    for _ in range(3):
        ops.conv2d(ga, wb, y, 3, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(ma, 'elu'), cin=4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d(ga, wb, y, 3, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(ma, 'elu'), cin=4)
    e1.record(); torch.cuda.synchronize()
    print('B%d %d^2 1->%d path %d  max err %.2e  %.1f us' % (B, H, Cout, path, err, e0.elapsed_time(e1) / 20 * 1e3))
