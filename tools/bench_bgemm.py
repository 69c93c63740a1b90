"""hv_bgemm_nt_h alone (fp16 x fp16 -> fp16, the attention block's A V product): python tools/bench_bgemm.py [M=1024] [N=1024] [K=1024] [batch=16] [iters=20]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan  # noqa: F401
from hvgan import lib
a = [int(v) for v in sys.argv[1:]]
M, N, K, batch, iters = (a + [1024, 1024, 1024, 16, 20][len(a):])[:5]
dev = torch.device('cuda:0')
rot = 3
As = [torch.randn(batch, M, K, device=dev).half() for _ in range(rot)]
Bs = [torch.randn(batch, N, K, device=dev).half() for _ in range(rot)]
Cs = [torch.zeros(batch, M, N, device=dev, dtype=torch.float16) for _ in range(rot)]
def run(i):
    lib.get().call('hv_bgemm_nt_h', lib.ptr(As[i % rot]), 1, K, ctypes.c_longlong(M * K), lib.ptr(Bs[i % rot]), 1, K, ctypes.c_longlong(N * K), lib.ptr(Cs[i % rot]), N,
                   ctypes.c_longlong(M * N), M, N, K, batch, ctypes.c_float(1.0), None, ctypes.c_longlong(0), 0, lib.stream())
for i in range(3):
    run(i)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(iters):
        run(i)
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / iters * 1e3
print('bgemm %dx%dx%d x%d DMA=%s VAR=%s: %.1f us  %.0f TF' % (M, N, K, batch, os.environ.get('HV_BGEMM_DMA', '1'), os.environ.get('HV_BGEMM_VAR', '0'), us, 2.0 * M * N * K * batch / us / 1e6))
