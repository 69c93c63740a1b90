"""What this box's HBM does for pure writes, pure reads and copies (torch kernels, 16-byte lanes, rotating 8 x 67 MB buffers): the roofs the write-bound thin kernels
are priced against (the guide's 8 TB/s is the read peak)."""
import torch
dev = torch.device('cuda:0')
n = 32 * 1024 * 1024      # halfs: 67 MB
bufs = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(8)]
outs = [torch.empty(n, dtype=torch.float16, device=dev) for _ in range(8)]
def timeit(f, iters=40):
    for i in range(4):
        f(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            f(i)
    g.replay(); torch.cuda.synchronize()
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
mb = n * 2 / 1e6
t = timeit(lambda i: bufs[i % 8].zero_())
print('fill  %6.1f us  %5.2f TB/s written' % (t, mb / t / 1e6 * 1e6 / 1e6))
t = timeit(lambda i: torch.sum(bufs[i % 8], dtype=torch.float32))
print('sum   %6.1f us  %5.2f TB/s read' % (t, mb / t))
t = timeit(lambda i: outs[i % 8].copy_(bufs[i % 8]))
print('copy  %6.1f us  %5.2f TB/s read + written' % (t, 2 * mb / t))
t = timeit(lambda i: torch.add(bufs[i % 8], 1.0, out=outs[i % 8]))
print('add   %6.1f us  %5.2f TB/s read + written' % (t, 2 * mb / t))
