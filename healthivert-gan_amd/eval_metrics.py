"""In-training evaluation on the device (reference train.py:37-48 dice_score / iou_score, :50-160 evaluate_model; SURVEY.md
section 8f row f3).

The reference evaluates sample by sample on the host: every sample's images go device -> numpy for skimage's SSIM / PSNR and
back through torch.tensor for Dice / IoU.  Here the eval forward, the SHRM compositing, the thresholds and the five metrics of a
whole batch stay on the device (`evaluate_batch`), and `evaluate_model` -- same signature and return tuple as the reference's --
reads back one [N,5] table at the end.  The image dump (torchvision.utils.save_image, train.py:148-159) is visualisation and is
left to the caller (`return_images=True` hands over the tensors the reference would tile).
"""
import ctypes

import torch

from . import lib as _lib
from . import ops
from .lib import ptr, stream


def dice_score(pred, target, smooth=1e-5):
    """train.py:37-41, same arithmetic on whatever device the tensors live on."""
    p, t = pred.contiguous().view(-1), target.contiguous().view(-1)
    inter = (p * t).sum()
    return (2. * inter + smooth) / (p.sum() + t.sum() + smooth)


def iou_score(pred, target, smooth=1e-5):
    """train.py:43-48."""
    p, t = pred.contiguous().view(-1), target.contiguous().view(-1)
    inter = (p * t).sum()
    return (inter + smooth) / (p.sum() + t.sum() - inter + smooth)


@torch.no_grad()
def batch_metrics(stage2, fine_seg, coarse_seg, pred2, real_B, real_B_mask, normal_vert, mask, height, x1, x2, maxheight):
    """Generator outputs + batch attributes (device tensors, (B,1,H,W) images) -> (metrics [B,5] fp32 on the device: ssim, psnr,
    dice(coarse, normal_vert), iou(fine, label), diff_h %; inpainted, coarse_bin, fine_bin).  No host synchronisation."""
    L = _lib.get()
    _lib.require_gpu(stage2, fine_seg, coarse_seg, real_B, real_B_mask, normal_vert, mask)
    B, _, H, W = stage2.shape
    dev = stage2.device
    n = ctypes.c_longlong(B * H * W)
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    i64 = lambda t: t.to(dev, torch.int64).contiguous()
    stage2, fine_seg, coarse_seg, real_B, real_B_mask, normal_vert, mask = map(f32, (stage2, fine_seg, coarse_seg, real_B, real_B_mask, normal_vert, mask))
    height, x1, x2 = i64(height), i64(x1), i64(x2)
    # pred_h = pred2.T * maxheight (train.py:74): fp32 * int64 -> fp32, per sample
    pred_h = torch.empty(B, dtype=torch.float32, device=dev)
    mh = f32(maxheight)
    L.call('hv_affine', ptr(pred_h), ptr(f32(pred2).view(-1)), ctypes.c_longlong(B), ctypes.c_float(1.0), ctypes.c_float(0.0), stream())
    L.call('hv_mul', ptr(pred_h), ptr(mh), ctypes.c_longlong(B), stream())
    inpainted = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
    L.call('hv_shrm_composite', ptr(stage2), ptr(real_B), ptr(pred_h), ptr(height), ptr(x1), ptr(x2), ptr(inpainted), None, B, H, W, stream())
    cb, fb = torch.empty_like(coarse_seg), torch.empty_like(fine_seg)
    L.call('hv_threshold', ptr(coarse_seg), ptr(cb), n, ctypes.c_float(0.5), ctypes.c_float(1.0), stream())
    L.call('hv_threshold', ptr(fine_seg), ptr(fb), n, ctypes.c_float(0.5), ctypes.c_float(1.0), stream())
    out = torch.empty(B, 5, dtype=torch.float32, device=dev)
    need = L.size('hv_eval_metrics_workspace_bytes', B, H, W)
    ws, _ = ops._ws(need, dev, slot=4)
    L.call('hv_eval_metrics', ptr(inpainted), ptr(real_B), ptr(mask), ptr(cb), ptr(normal_vert), ptr(fb), ptr(real_B_mask), ptr(pred_h), ptr(height),
           B, H, W, ptr(out), ptr(ws), ctypes.c_size_t(ws.numel()), stream())
    return out, inpainted, cb, fb


@torch.no_grad()
def evaluate_batch(model, batch):
    """One test batch through model.set_input + the eval-mode generator (train.py:56-75) + batch_metrics."""
    model.set_input(batch)
    o = model.netG(model.real_A, model.mask, _one_minus(model.CAM), model.slice_ratio)          # CAM_temp = 1 - CAMs (train.py:71)
    coarse_seg, fine_seg, _, stage2, _, _, pred2 = o
    return batch_metrics(stage2, fine_seg, coarse_seg, pred2, model.real_B, model.real_B_mask, model.normal_vert, model.mask, model.height,
                         model.x1, model.x2, model.maxheight)


def _one_minus(t):
    out = torch.empty_like(t)
    _lib.get().call('hv_affine', ptr(out), ptr(t.contiguous()), ctypes.c_longlong(t.numel()), ctypes.c_float(-1.0), ctypes.c_float(1.0), stream())
    return out


@torch.no_grad()
def evaluate_model(model, test_loader, device=None, checkpoint_path=None, iteration=0, return_images=False):
    """Reference signature (train.py:50): -> (avg_ssim, avg_psnr, avg_dice, avg_iou, avg_diffh) over every sample of the loader.
    The model is put in eval mode and back in train mode like the reference; per-sample values stay on the device until the end."""
    model.eval()
    tables, last = [], None
    for batch in test_loader:
        m, inpainted, cb, fb = evaluate_batch(model, batch)
        tables.append(m)
        last = (inpainted, cb, fb)
    model.train()
    if not tables:
        nan = float('nan')
        return (nan,) * 5
    t = torch.cat(tables).double().mean(dim=0).cpu()
    res = tuple(float(v) for v in t)
    return (res, last) if return_images else res
