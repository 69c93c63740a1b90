"""Stand-alone duration of each phase hipGraph of the data-parallel step (one rank, HV_DDP_FORCE=1), replayed alone on an idle GPU, next to the
overlapped step time: which phases carry the step, and what the stream overlap buys.

    python tools/phase_graph_times.py [batch=16]
"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HV_PRECISION', 'fp16')
os.environ['HV_DDP_FORCE'] = '1'
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
import torch
import torch.distributed as dist
import bench
import hvgan  # noqa: F401
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
torch.manual_seed(1234)
opt = bench.make_opt('fp16')
m = Pix2PixModel(opt)
m.setup(opt)
m.set_input(synth.make_batch(B, 256, seed=1234))
for _ in range(m.GRAPH_WARMUP + 3):
    m.optimize_parameters()
torch.cuda.synchronize()
assert m._dp_graphs, 'phases were not captured'
N = 10
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    m.optimize_parameters()
e1.record()
torch.cuda.synchronize()
print('overlapped step: %.2f ms' % (e0.elapsed_time(e1) / N))
if os.environ.get('PHASE'):      # replay ONE phase graph over and over (for a kernel trace of that phase alone: tools/trace_gaps.py ... dense:<ms>)
    g = m._dp_graphs[os.environ['PHASE']]
    torch.cuda.synchronize()
    for _ in range(30):
        g.replay()
    torch.cuda.synchronize()
    dist.destroy_process_group()
    sys.exit(0)
if os.environ.get('ONLY_STEPS') == '1':       # for a kernel trace of the overlapped schedule (tools/trace_gaps.py ... dense:<ms>)
    dist.destroy_process_group()
    sys.exit(0)
import time
def _events_only(flat, after=None):
    st = m.grad_sync.exchange_stream(flat.device)
    st.wait_stream(after if after is not None else torch.cuda.current_stream(flat.device))
    with torch.cuda.stream(st):
        ev = torch.cuda.Event()
        ev.record(st)
    return ev


def _no_affine(flat, after=None):
    st = m.grad_sync.exchange_stream(flat.device)
    st.wait_stream(after if after is not None else torch.cuda.current_stream(flat.device))
    with torch.cuda.stream(st):
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)
        work.wait()
        ev = torch.cuda.Event()
        ev.record(st)
    return ev


for label, patch in (('with the exchange', None), ('stream waits + event only', _events_only), ('all_reduce without the pre-scale', _no_affine),
                     ('exchange calls skipped (one rank: the mean is the identity)', lambda flat, after=None: None)):
    if patch is not None:
        m.grad_sync.reduce = patch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    for _ in range(N):
        m.optimize_parameters()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print('%s: host enqueue %.2f ms/step, device %.2f ms/step' % (label, (t1 - t0) / N * 1e3, e0.elapsed_time(e1) / N))
order = ['real1', 'real2', 'real3', 'gfwd', 'fake1', 'fake2', 'fake3', 'dstep1', 'dstep2', 'dstep3', 'gbwd', 'gadam']
tot = 0.0
for name in order:
    g = m._dp_graphs[name]
    ms = 0.0
    for _ in range(N):
        torch.cuda.synchronize()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ms += e0.elapsed_time(e1)
    tot += ms / N
    print('%-8s %.3f ms' % (name, ms / N))
print('sum of the phases alone: %.2f ms' % tot)
dist.destroy_process_group()
