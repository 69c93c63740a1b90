"""Test-side helpers (CPU<->device plumbing with torch; never used by the product path)."""
import torch

import hvgan
from hvgan import ops


def dev():
    return torch.device('cuda:0')


def st(prec):
    """Storage dtype of activation tensors for a compute precision (fp16 mode: fp16 storage)."""
    return torch.float16 if prec in ('fp16', 'f16') else torch.float32


def to_act(x_nchw, CP=None, dtype=torch.float32):
    """CPU (B,C,H,W) -> device NHWC Act with channel stride CP (zero padded), stored as `dtype`."""
    B, C, H, W = x_nchw.shape
    CP = C if CP is None else CP
    t = torch.zeros(B, H, W, CP)
    t[..., :C] = x_nchw.permute(0, 2, 3, 1)
    return ops.Act(t.to(dev()).to(dtype).contiguous(), C, 0)


def from_act(a):
    """device Act -> CPU (B,C,H,W)."""
    return a.t[..., a.coff:a.coff + a.C].permute(0, 3, 1, 2).contiguous().float().cpu()


def ohwi(w, CinP=None, CoutF=None):
    """CPU [Cout,Cin,kh,kw] -> device [CoutF][kh*kw][CinP] (zero padded)."""
    Co, Ci, kh, kw = w.shape
    CinP = Ci if CinP is None else CinP
    CoutF = Co if CoutF is None else CoutF
    o = torch.zeros(CoutF, kh * kw, CinP)
    o[:Co, :, :Ci] = w.permute(0, 2, 3, 1).reshape(Co, kh * kw, Ci)
    return o.to(dev()).contiguous()


def ohwi_T(w, CoutP=None, CinB=None):
    """CPU [Cout,Cin,kh,kw] -> device data-gradient layout [CinB][kh*kw][CoutP]."""
    Co, Ci, kh, kw = w.shape
    CoutP = Co if CoutP is None else CoutP
    CinB = Ci if CinB is None else CinB
    o = torch.zeros(CinB, kh * kw, CoutP)
    o[:Ci, :, :Co] = w.permute(1, 2, 3, 0).reshape(Ci, kh * kw, Co)
    return o.to(dev()).contiguous()


def maxerr(a, b):
    return (a.double() - b.double()).abs().max().item()
