"""Inference-side tensor work of the reference's eval_3d_sagittal_twostage.py (run_model :96-130, load_model :32-38)
on the HIP path: eval-mode generator forward, pred_h = ceil(pred2_h*maxheight), SHRM re-compositing of the CT
((x+1)*127.5) and of the label map (fine_seg > 0.5 -> vert_id), all on the device, batched over slices.

The reference runs ~130 sequential bs=1 forwards per volume, each with a PIL->tensor->H2D hop and a D2H copy
(SURVEY.md section 3.2); z-slices of one stage are independent, so `synthesize` takes a whole batch of slices.
File I/O, connected-component cleaning and the uint8 mask-band construction stay on the caller's side.
"""
import ctypes

import torch

from . import lib as _lib
from . import ops
from .lib import ptr, stream


def load_generator(model_path, netG_params, device):
    """load_model of the reference (:32-38): Generator(cfg, True) + state dict, eval mode, on `device`."""
    from .models.inpaint_networks import Generator
    import os
    model = Generator(netG_params, True)
    if os.path.exists(model_path):
        model.load_state_dict(torch.load(model_path, map_location='cpu'))
        model.eval()
    model.to(device)
    return model


@torch.no_grad()
def synthesize(model, ct_masked, mask, cam, index_ratio, ori_ct, label, x1, x2, height, vert_id, maxheight=40):
    """Batched tensor part of run_model.

    ct_masked, mask, cam, ori_ct: (B,1,H,W) fp32 device tensors ([-1,1] for CT, [0,1] for mask/CAM); label: (B,1,H,W)
    fp32 label map; index_ratio: (B,) fp64; x1, x2, height: (B,) int64 device tensors; vert_id: python number.
    Returns (label_fake (B,H,W), ct_fake (B,H,W) in [0,255], pred_h_raw (B,)) -- all device tensors, no host sync."""
    L = _lib.get()
    _lib.require_gpu(ct_masked, mask, cam, ori_ct, label)
    B, _, H, W = ct_masked.shape
    dev = ct_masked.device
    n = ctypes.c_longlong(B * H * W)
    cam_t = torch.empty_like(cam)
    L.call('hv_affine', ptr(cam_t), ptr(cam.contiguous()), n, ctypes.c_float(-1.0), ctypes.c_float(1.0), stream())
    P = model.run_forward(ct_masked, mask, cam_t, index_ratio, training=False)
    pred = torch.empty(B, device=dev)
    L.call('hv_affine', ptr(pred), ptr(P.pred2), ctypes.c_longlong(B), ctypes.c_float(float(maxheight)), ctypes.c_float(0.0), stream())
    x1, x2, height = (t.to(dev).long().contiguous() for t in (x1, x2, height))
    ct = torch.empty(B, H, W, device=dev)
    L.call('hv_shrm_composite', ptr(P.x_stage2), ptr(ori_ct.contiguous()), ptr(pred), ptr(height), ptr(x1), ptr(x2), ptr(ct), None, B, H, W, stream())
    L.call('hv_affine', ptr(ct), ptr(ct), n, ctypes.c_float(127.5), ctypes.c_float(127.5), stream())
    seg = torch.empty(B, 1, H, W, device=dev)
    L.call('hv_threshold', ptr(P.fine_seg), ptr(seg), n, ctypes.c_float(0.5), ctypes.c_float(float(vert_id)), stream())
    lab = torch.empty(B, H, W, device=dev)
    L.call('hv_shrm_composite', ptr(seg), ptr(label.contiguous()), ptr(pred), ptr(height), ptr(x1), ptr(x2), ptr(lab), None, B, H, W, stream())
    return lab, ct, P.pred2.view(B).clone()
