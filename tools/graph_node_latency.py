"""What does a kernel node cost inside a replayed hipGraph on this runtime, and what do parallel branches cost?  N kernels per branch (hv_affine over n floats),
1 / 2 / 3 independent branches forked from the capture stream and joined at the end; ideal = the time of ONE branch when the kernels are small enough to
run side by side.     python tools/graph_node_latency.py [N=100]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan  # noqa: F401
from hvgan import lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device('cuda:0')
L = lib.get()
def k(t):
    L.call('hv_affine', lib.ptr(t), lib.ptr(t), ctypes.c_longlong(t.numel()), ctypes.c_float(1.0), ctypes.c_float(0.0), lib.stream())
def timeit(g, reps=10):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
cap = torch.cuda.Stream()
sides = [torch.cuda.Stream() for _ in range(3)]
for n in (1024, 1 << 20, 1 << 23):
    bufs = [torch.zeros(n, device=dev) for _ in range(3)]
    k(bufs[0]); torch.cuda.synchronize()
    res = []
    for nb in (1, 2, 3):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=cap):
            cur = torch.cuda.current_stream()
            for b in range(1, nb):
                sides[b].wait_stream(cur)
                with torch.cuda.stream(sides[b]):
                    for _ in range(N):
                        k(bufs[b])
            for _ in range(N):
                k(bufs[0])
            for b in range(1, nb):
                cur.wait_stream(sides[b])
        res.append(timeit(g))
    print('n = %8d floats, %d kernels per branch: 1 branch %.1f us (%.2f per kernel), 2 branches %.1f us (%.2fx), 3 branches %.1f us (%.2fx)'
          % (n, N, res[0], res[0] / N, res[1], res[1] / res[0], res[2], res[2] / res[0]))

# ---- the alternative to branches inside one graph: one chain graph per stream, launched side by side
for n in (1024, 1 << 20, 1 << 23):
    bufs = [torch.zeros(n, device=dev) for _ in range(3)]
    graphs = []
    for b in range(3):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=sides[b]):
            for _ in range(N):
                k(bufs[b])
        graphs.append(g)
    def run(nb):
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event(); ev.record(main)
        evs = []
        for b in range(nb):
            sides[b].wait_event(ev)
            with torch.cuda.stream(sides[b]):
                graphs[b].replay()
                e = torch.cuda.Event(); e.record(sides[b]); evs.append(e)
        for e in evs:
            main.wait_event(e)
    res = []
    for nb in (1, 2, 3):
        run(nb); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(nb)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 10 * 1e3)
    print('n = %8d floats, one chain graph of %d kernels per stream: 1 stream %.1f us, 2 streams %.1f us (%.2fx), 3 streams %.1f us (%.2fx)'
          % (n, N, res[0], res[1], res[1] / res[0], res[2], res[2] / res[0]))

# ---- is a branchy region executed level by level?  branch A: 100 tiny kernels; branch B: ONE long kernel (hv_affine over 64 M floats, ~400 us)
big = torch.zeros(1 << 26, device=dev)
small = torch.zeros(1024, device=dev)
k(big); torch.cuda.synchronize()
cap_stream = torch.cuda.Stream()
def only_a():
    for _ in range(N): k(small)
def only_b():
    k(big)
def both(first_long):
    cur = torch.cuda.current_stream()
    sides[1].wait_stream(cur)
    with torch.cuda.stream(sides[1]):
        if first_long: k(big)
        else:
            for _ in range(N): k(small)
    if first_long:
        for _ in range(N): k(small)
    else: k(big)
    cur.wait_stream(sides[1])
for name, fn in (('100 tiny alone', only_a), ('1 long alone', only_b), ('long on the side branch, tiny on the main branch', lambda: both(True)), ('tiny on the side branch, long on the main branch', lambda: both(False))):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap_stream):
        fn()
    print('%-52s %.1f us' % (name, timeit(g)))

# ---- does a fork-join region tax the nodes outside it?  chain of 50 | fork: 2 branches x 25 | join | chain of 50  (150 nodes on the critical path)
def mixed():
    cur = torch.cuda.current_stream()
    for _ in range(50): k(small)
    sides[1].wait_stream(cur)
    with torch.cuda.stream(sides[1]):
        for _ in range(25): k(bufs[1])
    for _ in range(25): k(small)
    cur.wait_stream(sides[1])
    for _ in range(50): k(small)
def chain125():
    for _ in range(125): k(small)
bufs = [torch.zeros(1024, device=dev) for _ in range(3)]
for name, fn in (('chain of 125 tiny', chain125), ('50 chain | 2 x 25 in a fork-join | 50 chain', mixed)):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap_stream):
        fn()
    print('%-52s %.1f us' % (name, timeit(g)))
