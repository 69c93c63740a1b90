"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI.

The reference has no distributed code (only nn.DataParallel on the discriminators, reference
models/networks.py:112-116).  Here every rank runs the whole step on its own 16-slice batch (all of the
reference's per-batch quirks stay per-rank) and the four networks' gradients are averaged with one flat
all-reduce each, issued on a side HIP stream so that D_k's reduction overlaps D_{k+1}'s forward/backward
(SURVEY.md section 8e).  With no process group initialised every call is a no-op.
"""
import os

import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, group=None):
        self.group = group
        self.stream = None
        self.pending = []

    @staticmethod
    def active():
        # HV_DDP_FORCE=1: run the exchange even in a one-rank group (RCCL smoke test of the exact call sequence on a single GPU)
        return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get('HV_DDP_FORCE') == '1')

    def reduce(self, flat):
        """Start averaging `flat` (a network's flat gradient buffer) across ranks."""
        if not self.active():
            return
        ws = dist.get_world_size(self.group)
        if flat.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=flat.device)
            self.stream.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self.stream):
                flat.mul_(1.0 / ws)
                work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self.pending.append((work, flat))
        else:   # gloo on CPU tensors (tests)
            flat.mul_(1.0 / ws)
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)

    def wait(self):
        """Make the current stream wait for every outstanding reduction (before the optimiser reads the gradients)."""
        for work, flat in self.pending:
            work.wait()
            if flat.is_cuda:
                torch.cuda.current_stream(flat.device).wait_stream(self.stream)
        self.pending = []


def broadcast_parameters(modules, src=0, group=None):
    """Make every rank start from rank `src`'s weights and buffers."""
    if not GradSync.active():
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src=src, group=group)
