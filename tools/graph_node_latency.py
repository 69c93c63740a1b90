"""What does one dependent kernel node cost inside a replayed hipGraph on this runtime?  Chains of N tiny kernels (hv_affine over 1 K floats: ~2 us of work)
captured (a) on one stream, (b) ping-ponging between two streams (every node joins across streams), (c) as two independent chains on two streams.
    python tools/graph_node_latency.py [N=200]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan  # noqa: F401
from hvgan import lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device('cuda:0')
x = torch.zeros(1024, device=dev)
y = torch.zeros(1024, device=dev)
L = lib.get()
def k(t):
    L.call('hv_affine', lib.ptr(t), lib.ptr(t), ctypes.c_longlong(t.numel()), ctypes.c_float(1.0), ctypes.c_float(0.0), lib.stream())
def timeit(g, reps=20):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
k(x); torch.cuda.synchronize()
cap = torch.cuda.Stream()
side = torch.cuda.Stream()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=cap):
    for _ in range(N):
        k(x)
t1 = timeit(g1)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2, stream=cap):
    cur = torch.cuda.current_stream()
    for i in range(N):
        if i & 1:
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                k(x)
            cur.wait_stream(side)
        else:
            k(x)
t2 = timeit(g2)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3, stream=cap):
    cur = torch.cuda.current_stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(N):
            k(y)
    for _ in range(N):
        k(x)
    cur.wait_stream(side)
t3 = timeit(g3)
# eager chain for comparison
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    k(x)
e1.record(); torch.cuda.synchronize()
te = e0.elapsed_time(e1) * 1e3
print('graph replay, %d tiny dependent kernels: one stream %.2f us/node; alternating two streams %.2f us/node; two independent chains of %d: %.2f us per node pair; eager one stream %.2f us/launch'
      % (N, t1 / N, t2 / N, N, t3 / N, te / N))
