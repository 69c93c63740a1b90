"""logits_dgrad_kernel (PatchGAN logits layer's data gradient: 4x4 s1 p1, 1 -> Cout channels) against torch CPU + time; HV_LOGITS_DGRAD=0 is the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import hvgan
from hvgan import ops, lib
dev = torch.device('cuda:0')
for (B, H, Cout, live) in [(2, 6, 32, 1), (2, 17, 64, 3), (3, 30, 512, 1), (16, 30, 512, 1), (16, 62, 512, 1)]:
    g_ = torch.Generator().manual_seed(Cout + H)
    gy = torch.zeros(B, 4, H, H); gy[:, :live] = torch.randn(B, live, H, H, generator=g_)
    w = torch.randn(live, Cout, 4, 4, generator=g_) / 4.0                   # forward conv Cout -> live
    Ho = H + 1
    m = torch.randn(B, Cout, Ho, Ho, generator=g_)
    ga = ops.Act(gy.permute(0, 2, 3, 1).contiguous().to(dev).half(), 4, 0)
    ma = ops.Act(m.permute(0, 2, 3, 1).contiguous().to(dev).half())
    wb = torch.zeros(Cout, 16, 4); wb[:, :, :live] = w.reshape(live, Cout, 16).permute(1, 2, 0)   # [ci = Cout rows][taps][coP = 4]
    wb = wb.to(dev)
    for acc in (0, 1):
        y = ops.Act(torch.full((B, Ho, Ho, Cout), 0.25, device=dev, dtype=torch.float16))
        ops.conv2d(ga, wb, y, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(ma, 'lrelu'), accumulate=acc, cin=4)
        path = lib.get().size('hv_last_kernel_path')
        torch.cuda.synchronize()
        ref = F.conv_transpose2d(gy[:, :live].half().float(), w.half().float(), None, stride=1, padding=1)
        mh = m.half().float()
        want = 0.25 * acc + ref * torch.where(mh > 0, torch.ones_like(mh), torch.full_like(mh, 0.2))
        err = (y.t.float().cpu().permute(0, 3, 1, 2) - want).abs().max().item()
        print('B%d %d^2 %d->%d acc %d path %d  max err %.2e (scale %.2f)' % (B, H, live, Cout, acc, path, err, want.abs().max().item()))
    for _ in range(3):
        ops.conv2d(ga, wb, y, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(ma, 'lrelu'), cin=4)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d(ga, wb, y, 4, 1, 1, 1, transposed=True, precision='fp16', w_h=wb.half(), mul=(ma, 'lrelu'), cin=4)
    e1.record(); torch.cuda.synchronize()
    print('   %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3))
