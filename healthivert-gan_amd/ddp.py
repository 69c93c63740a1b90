"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code (only nn.DataParallel on the discriminators, reference
models/networks.py:112-116).  Here every rank runs the whole step on its own 16-slice batch (all of the
reference's per-batch quirks stay per-rank) and the four networks' gradients are averaged with one flat
all-reduce each on a dedicated HIP stream:

* every collective of the step goes to ONE communicator on ONE stream (the exchange stream) in the fixed order D_1, D_2, D_3, G -- the same order
  on every rank by construction (RCCL requires it);
* `HV_DP_SCHEDULE=captured`: the collectives INSIDE the step's one hipGraph, as one exchange BRANCH of it (`reduce_branch`): the branch waits for
  D_k's stream where D_k's gradients are final, runs `ncclAllReduce(..., ncclAvg)`, and only D_k's optimiser step waits for it -- the other
  discriminators' passes run beside it; the generator's reduction sits between its backward and its Adam step on the same branch;
* `HV_DP_SCHEDULE=graphs`: the step cut into its three hipGraphs where the exchanges belong, the same collectives issued eagerly between them
  (`reduce`) -- the fallback when a runtime refuses to capture RCCL kernels, and the only schedule for gloo (tests, rehearsals);
* with an RCCL process group BOTH schedules talk to RCCL directly through `RcclComm` below -- a communicator of our own (ncclGetUniqueId on rank 0,
  broadcast through the process group, ncclCommInitRank) on torch's own librccl.so.  torch's ProcessGroupNCCL is used for the rendezvous, the weight
  broadcast and the bench clock only: its watchdog thread polls the end events of its collectives, and a poll that lands while a stream of the
  process is capturing aborts the process on this stack (`hipErrorCapturedEvent`), so nothing of the step is issued through it;
* which schedule a multi-GPU job takes is decided by a preflight on the job's own first batch (`Pix2PixModel.dp_preflight`): both are run, their
  results compared across the ranks, the faster correct one is kept.

`init_from_env()` joins the process group that `python -m torch.distributed.run` describes in the environment, so
a reference `train.py` needs no edit.  With no process group initialised every call is a no-op.
"""
import ctypes
import os

import torch
import torch.distributed as dist


def init_from_env():
    """Join the job described by RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (set by torch.distributed.run) if there is one and
    nobody has joined it yet.  Returns the local device index this rank must drive, or None when the process is not part
    of a multi-process job.  Backend: RCCL ('nccl' IS RCCL on ROCm); HV_DDP_BACKEND=gloo for dry runs / tests."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    force = os.environ.get('HV_DDP_FORCE') == '1'       # a one-rank group: the exact data-parallel schedule (phase graphs, exchange stream, RCCL calls) on one GPU
    if (world <= 1 and not force) or not dist.is_available():
        return None
    if world <= 1:
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('LOCAL_RANK', '0')
        world = 1
    local = int(os.environ.get('LOCAL_RANK', os.environ.get('RANK', '0')))
    ndev = torch.cuda.device_count()
    if ndev and local >= ndev:      # rehearsal of N ranks on fewer devices (gloo backend; RCCL refuses two ranks on one device)
        local %= ndev
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:      # (no launcher: a one-rank rehearsal) any free port of this host
            import socket
            with socket.socket() as sk:
                sk.bind(('127.0.0.1', 0))
                os.environ['MASTER_PORT'] = str(sk.getsockname()[1])
        backend = os.environ.get('HV_DDP_BACKEND', 'nccl')
        kw = {}
        if backend == 'nccl':
            torch.cuda.set_device(local)
            kw['device_id'] = torch.device('cuda', local)
        dist.init_process_group(backend, rank=int(os.environ['RANK']), world_size=world, **kw)
    return local


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


class RcclComm:
    """A communicator opened on librccl directly (ctypes): ncclAllReduce on a stream of the caller's choice, capturable into a hipGraph.  The library is
    the one torch itself loaded (torch/lib/librccl.so), the rendezvous goes through the existing process group (any backend)."""
    NCCL_FLOAT, NCCL_AVG = 7, 4        # rccl.h: ncclFloat32, ncclAvg

    class _UniqueId(ctypes.Structure):
        _fields_ = [('internal', ctypes.c_byte * 128)]

    def __init__(self, group=None):
        path = os.path.join(os.path.dirname(torch.__file__), 'lib', 'librccl.so')
        self.lib = ctypes.CDLL(path if os.path.exists(path) else 'librccl.so')
        self.lib.ncclGetErrorString.restype = ctypes.c_char_p
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        uid = self._UniqueId()
        if rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), 'ncclGetUniqueId')
        box = [bytes(uid.internal)]
        dist.broadcast_object_list(box, src=0, group=group)
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        self.lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._UniqueId, ctypes.c_int]
        self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank), 'ncclCommInitRank')
        self.lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
        self.lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        self.world = world
        import atexit
        atexit.register(self.close)

    def close(self):
        """ncclCommDestroy (idempotent; registered with atexit so that teardown does not leave a live communicator behind).  The device is drained
        first: a collective still in flight on another stream would otherwise be destroyed under its kernel."""
        comm, self.comm = self.comm, None
        if comm is not None and comm.value:
            try:
                torch.cuda.synchronize()
                self.lib.ncclCommDestroy(comm)
            except Exception:      # noqa: BLE001 -- interpreter teardown: the runtime may already be gone
                pass

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError('%s failed: %s' % (what, (self.lib.ncclGetErrorString(rc) or b'?').decode()))

    def all_reduce_mean(self, flat):
        """flat (fp32, device) <- mean over the ranks, in place, on the current stream."""
        if self.comm is None:
            raise RuntimeError('RcclComm: the communicator has been closed')
        self._check(self.lib.ncclAllReduce(flat.data_ptr(), flat.data_ptr(), flat.numel(), self.NCCL_FLOAT, self.NCCL_AVG, self.comm,
                                           torch.cuda.current_stream(flat.device).cuda_stream), 'ncclAllReduce')


class GradSync:
    def __init__(self, group=None):
        self.group = group
        self.stream = None
        self.rccl = None
        self._chain = {}

    @staticmethod
    def active():
        # HV_DDP_FORCE=1: run the exchange even in a one-rank group (RCCL smoke test of the exact call sequence on a single GPU)
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('HV_DDP_FORCE') == '1')

    def capturable(self):
        """True when the transport's collectives are stream-ordered device work that a hipGraph capture records (RCCL); gloo's run on the host."""
        return self.active() and 'nccl' in str(dist.get_backend(self.group))

    def open(self):
        """Open the RCCL communicator (RCCL process groups only; a collective over the process group: every rank calls it at the same point -- the
        first step's exchange -- and never under stream capture)."""
        if self.rccl is None and self.capturable():
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('GradSync: the RCCL communicator must be opened before the capture (run an eager step first)')
            self.rccl = RcclComm(self.group)
        return self.rccl

    def close(self):
        if self.rccl is not None:
            self.rccl.close()
            self.rccl = None

    def exchange_stream(self, device):
        if self.stream is None:
            from . import engine
            # default priority.  A high-priority exchange stream (HV_DDP_COMM_PRIO=-1) gets a hardware queue of its own whose scheduling
            # slows the compute streams' phase graphs: one-rank rehearsal of the exact schedule, same device, 13.85 ms/step against 11.73
            self.stream = engine.named_stream('exchange', device, priority=int(os.environ.get('HV_DDP_COMM_PRIO', '0')))
        return self.stream

    def reduce_branch(self, flat, producer=None):
        """Captured schedule: average `flat` across the ranks, as ONE chain of collectives over the whole step -- every ncclAllReduce of the step is ordered
        behind the previous one by an explicit edge, in the order of the calls (D_1, D_2, D_3, G: the same on every rank by construction; RCCL requires
        it) -- ordered after everything queued on `producer` (default: the current stream), which continues when the mean is there.  Under stream capture
        the collectives are kernel nodes of the graph being captured, and the other streams keep running beside them.
        Two forms (HV_DP_BRANCH): 'chain' (default) issues the collective on the PRODUCER's stream behind an event recorded after the previous collective
        (D_k's mean beside the other discriminators' passes; the only extra edges are joins between branches that exist anyway); 'stream' issues all of
        them on the exchange stream (a branch of its own: hipStreamEndCapture of ROCm 7.0 / 7.2 crashes on that topology, kept for newer runtimes)."""
        if not flat.is_cuda or flat.dtype != torch.float32:
            raise RuntimeError('reduce_branch: fp32 device tensors only')
        comm = self.open()
        if comm is None:
            raise RuntimeError('reduce_branch: needs an RCCL process group')
        producer = producer if producer is not None else torch.cuda.current_stream(flat.device)
        if os.environ.get('HV_DP_BRANCH', 'chain') == 'stream':
            st = self.exchange_stream(flat.device)
            st.wait_stream(producer)
            with torch.cuda.stream(st):
                comm.all_reduce_mean(flat)
            producer.wait_stream(st)
            return
        capturing = torch.cuda.is_current_stream_capturing()
        prev = self._chain.get(capturing)
        if prev is not None and prev[0] != producer.cuda_stream:
            producer.wait_event(prev[1])
        with torch.cuda.stream(producer):
            comm.all_reduce_mean(flat)
            ev = torch.cuda.Event()
            ev.record(producer)
        self._chain[capturing] = (producer.cuda_stream, ev)

    def chain_reset(self):
        """Start of a step (or of a capture): the first collective has no predecessor inside it (an event recorded outside a capture must not be waited
        for inside it, and the previous step's collectives are ordered before this step's by the streams themselves)."""
        self._chain = {}

    def reduce(self, flat, after=None):
        """Cut schedule: average `flat` (a network's flat gradient buffer) across ranks on the exchange stream, ordered after everything queued
        on stream `after` (default: the current stream).  Returns an event that is complete when `flat` holds the mean (None
        for CPU tensors, which are reduced synchronously); the consumer's stream waits for it -- nothing blocks the host."""
        if not self.active():
            return None
        ws = dist.get_world_size(self.group)
        if not flat.is_cuda:   # gloo on CPU tensors (tests, dry runs)
            flat.mul_(1.0 / ws)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return None
        from . import lib as _lib
        st = self.exchange_stream(flat.device)
        st.wait_stream(after if after is not None else torch.cuda.current_stream(flat.device))
        with torch.cuda.stream(st):
            if self.capturable():
                # RCCL averages inside the collective (ncclAvg): no pre-scale pass over the buffer; through our own communicator, so that no
                # ProcessGroupNCCL work item (and no watchdog poll of its events) exists around the step's graph captures
                self.open().all_reduce_mean(flat)
            else:
                # gloo has no AVG: pre-scale (1/ws) with the library's own pointwise kernel, then SUM -- the mean of the ranks' gradients
                _lib.get().call('hv_affine', _lib.ptr(flat), _lib.ptr(flat), ctypes.c_longlong(flat.numel()), ctypes.c_float(1.0 / ws),
                                ctypes.c_float(0.0), _lib.stream())
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                work.wait()        # gloo: host wait
            ev = torch.cuda.Event()
            ev.record(st)
        return ev


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank `src`'s weights and buffers."""
    if not GradSync.active():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=group)
