"""Digest and timing of hv_ca_fuse (forward) on a random score matrix: run with HV_CA_FUSE_TILE=0 and =1, the digests must match bit for bit."""
import ctypes
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hvgan  # noqa: F401
from hvgan import lib

B, h, w = 16, 32, 32
L = h * w
g = torch.Generator().manual_seed(5)
S = torch.randn(B, L, L, generator=g).cuda()
out = torch.empty_like(S)
L_ = lib.get()
for adj in (0, 1):
    L_.call('hv_ca_fuse', lib.ptr(S), lib.ptr(out), B, h, w, adj, lib.stream())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        L_.call('hv_ca_fuse', lib.ptr(S), lib.ptr(out), B, h, w, adj, lib.stream())
    e1.record()
    torch.cuda.synchronize()
    o = out.double()
    print('adjoint %d: %.1f us  sum %.10e  abs %.10e  corner %.8e %.8e' % (adj, e0.elapsed_time(e1) / 20 * 1e3, o.sum().item(), o.abs().sum().item(),
                                                                  out[0, 0, 0].item(), out[B - 1, L - 1, L - 1].item()))
