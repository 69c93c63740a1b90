"""Device-side RHLV quantification (reference evaluation/RHLV_quantification.py), SURVEY.md section 8f row f4.

Same entry points as the reference module, on device tensors:
  calculate_rhlv(segmentation_fake, segmentation_label, center_z, length, vertebra, height_threshold)   (:120-147)
  rhlv_volume(vol_fake, vol_label, label_index, length_divisor, height_threshold)       per-vertebra body of :160-178
Volumes are [H, W, Z] tensors (any strides; float32 or uint8) already resident in HBM, e.g. the label volume infer.process_volume
just produced; the result is read back as five Python floats.  The file I/O / Excel part of the reference stays on the host.
"""
import ctypes

import torch

from . import lib as _lib
from . import ops

INT_MIN = -2147483648


def _prep(v, device):
    t = torch.as_tensor(v)
    if t.dtype not in (torch.float32, torch.uint8):
        t = t.to(torch.float32)
    return t.to(device)


def _run(fake, label, label_index, length_divisor, z_lo, z_hi, height_threshold):
    L = _lib.get()
    dev = fake.device if isinstance(fake, torch.Tensor) and fake.is_cuda else torch.device('cuda', torch.cuda.current_device())
    f, l = _prep(fake, dev), _prep(label, dev)
    if l.dtype != f.dtype:
        f, l = f.to(torch.float32), l.to(torch.float32)
    _lib.require_gpu(f, l)
    if f.shape != l.shape or f.dim() != 3 or f.stride() != l.stride():
        raise ValueError('rhlv: two [H, W, Z] volumes of equal shape and strides expected')
    H, W, Z = f.shape
    out = torch.zeros(14, dtype=torch.float64, device=dev)
    need = L.size('hv_rhlv_workspace_bytes', W, Z)
    ws, _ = ops._ws(need, dev, slot=3)
    L.call('hv_rhlv', _lib.ptr(f), _lib.ptr(l), 0 if f.dtype == torch.float32 else 1, ctypes.c_longlong(f.stride(0)), ctypes.c_longlong(f.stride(1)),
           ctypes.c_longlong(f.stride(2)), H, W, Z, ctypes.c_float(label_index), int(length_divisor), int(z_lo), int(z_hi),
           ctypes.c_double(height_threshold), _lib.ptr(out), _lib.ptr(ws), ctypes.c_size_t(ws.numel()), _lib.stream())
    return out


def calculate_rhlv(segmentation_fake, segmentation_label, center_z, length, vertebra=None, height_threshold=0.64):
    """-> (all_rhlv, pre_rhlv, mid_rhlv, post_rhlv, relative_height_label): binary volumes, slices [center_z-length, center_z+length)."""
    o = _run(segmentation_fake, segmentation_label, -1.0, 1, int(center_z) - int(length), int(center_z) + int(length), height_threshold).cpu()
    return tuple(float(v) for v in o[:5])


def rhlv_volume(vol_fake, vol_label, label_index, length_divisor=5, height_threshold=0.64, return_means=False):
    """Label volumes carrying vertebra ids -> the five values of calculate_rhlv for vertebra `label_index`, or None if the original
    volume does not contain it (the reference's `continue`)."""
    o = _run(vol_fake, vol_label, float(label_index), length_divisor, INT_MIN, 0, height_threshold).cpu()
    if o[13] == 0:
        return None
    res = tuple(float(v) for v in o[:5])
    return (res, [float(v) for v in o[5:13]]) if return_means else res
