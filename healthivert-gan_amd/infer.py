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
    # the reference synthesises slice by slice at batch 1: each slice's own mask band selects its valid attention patches
    P = model.run_forward(ct_masked, mask, cam_t, index_ratio, training=False, per_sample_mask=True)
    lab, ct = recomposite(P.x_stage2, P.fine_seg, P.pred2, ori_ct, label, x1, x2, height, vert_id, maxheight)
    return lab, ct, P.pred2.view(B).clone()


@torch.no_grad()
def recomposite(x_stage2, fine_seg, pred2, ori_ct, label, x1, x2, height, vert_id, maxheight=40):
    """Post-processing of run_model (reference :103-130) for a batch: pred_h = max(ceil(pred2 * maxheight), height), generated rows
    [x_upper, x_bottom) between the re-stacked original rows, CT back to [0, 255], label = (fine_seg > 0.5) * vert_id in the same rows.
    x_stage2, fine_seg, ori_ct, label: (B,1,H,W) fp32 device tensors; pred2: (B,1) or (B,).  Returns (label_fake, ct_fake) (B,H,W)."""
    L = _lib.get()
    _lib.require_gpu(x_stage2, fine_seg, ori_ct, label)
    B, _, H, W = x_stage2.shape
    dev = x_stage2.device
    n = ctypes.c_longlong(B * H * W)
    pred = torch.empty(B, device=dev)
    L.call('hv_affine', ptr(pred), ptr(pred2.contiguous()), ctypes.c_longlong(B), ctypes.c_float(float(maxheight)), ctypes.c_float(0.0), stream())
    x1, x2, height = (t.to(dev).long().contiguous() for t in (x1, x2, height))
    ct = torch.empty(B, H, W, device=dev)
    L.call('hv_shrm_composite', ptr(x_stage2.contiguous()), ptr(ori_ct.contiguous()), ptr(pred), ptr(height), ptr(x1), ptr(x2), ptr(ct), None, B, H, W, stream())
    L.call('hv_affine', ptr(ct), ptr(ct), n, ctypes.c_float(127.5), ctypes.c_float(127.5), stream())
    seg = torch.empty(B, 1, H, W, device=dev)
    L.call('hv_threshold', ptr(fine_seg.contiguous()), ptr(seg), n, ctypes.c_float(0.5), ctypes.c_float(float(vert_id)), stream())
    lab = torch.empty(B, H, W, device=dev)
    L.call('hv_shrm_composite', ptr(seg), ptr(label.contiguous()), ptr(pred), ptr(height), ptr(x1), ptr(x2), ptr(lab), None, B, H, W, stream())
    return lab, ct


# ------------------------------------------------------------------------------------------------ stage-batched volume driver
_POOL = None


def _pool():
    global _POOL
    if _POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) // 2)))
    return _POOL


def _clean_components(mask, min_size):
    """remove_small_connected_components of the reference (eval_3d_sagittal_twostage.py:15-30): 8-connectivity."""
    import numpy as np
    from scipy.ndimage import label as cc_label
    lab, n = cc_label(mask, np.ones((3, 3), dtype=np.int32))
    if n:
        small = np.bincount(lab.ravel(), minlength=n + 1) < min_size     # component sizes in one pass
        small[0] = False
        mask[small[lab]] = 0
    return mask


def prepare_slice(cam2d, label2d, ct2d, vert_id, maxheight=40):
    """CPU part of run_model (reference :46-98): bbox of the vertebra, 41-row mask band (note `max_x+1`, :75), masked CT and
    CAM re-stacked around the band, uint8 quantisation, ToTensor/Normalize.  Returns None when the vertebra is absent."""
    import numpy as np
    vl = np.zeros_like(label2d)
    vl[label2d == vert_id] = 1
    vl = _clean_components(vl, 50)
    coords = np.argwhere(vl)
    if coords.size == 0:
        return None
    x1, x2 = int(coords[:, 0].min()), int(coords[:, 0].max())
    width = vl.shape[0]
    height = x2 - x1
    if height > maxheight:
        x_mean = int(np.mean(coords[:, 0]))
        x1 = x_mean - 20
        x2 = x1 + 40
    mask_x, h2 = (x1 + x2) // 2, maxheight
    if mask_x <= h2 // 2:
        min_x = 0
    elif width - mask_x <= h2 / 2:
        min_x = width - h2
    else:
        min_x = mask_x - h2 // 2
    max_x = min_x + h2
    mask = np.zeros(vl.shape, np.uint8)
    mask[min_x:max_x + 1] = 255
    ct_u8 = ct2d.astype(np.uint8)
    ct_masked = np.zeros_like(mask)
    ct_masked[:min_x] = ct_u8[(x1 - min_x):x1]
    ct_masked[max_x:] = ct_u8[x2:x2 + (width - max_x)]
    cam = np.zeros_like(mask)
    cam_u8 = cam2d.astype(np.uint8)
    cam[:min_x] = cam_u8[(x1 - min_x):x1]
    cam[max_x:] = cam_u8[x2:x2 + (width - max_x)]
    f = lambda a, norm: ((a.astype(np.float32) / 255.0 - 0.5) / 0.5) if norm else a.astype(np.float32) / 255.0
    return dict(ct_masked=f(ct_masked, True), ori_ct=f(ct_u8, True), mask=f(mask, False), cam=f(cam, False), x1=x1, x2=x2, height=height)


def _stage(model, cam_vol, labels, cts, zs, ratios, vert_id, device, maxheight):
    """One synthesis stage for all z-slices at once: labels/cts are lists of 2-D arrays (per z), cam_vol is [Z, H, W].  Returns new lists
    (slices where the vertebra is absent pass through unchanged, like the reference's `output == None`)."""
    import numpy as np
    # host-side slice preparation (connected components, bounding box, band re-stacking) is independent per slice: a small thread
    # pool keeps it off the critical path of the three batched generator launches (numpy / scipy release the GIL in their loops)
    prep1 = lambda iz: prepare_slice(cam_vol[iz[1]], labels[iz[0]], cts[iz[0]], vert_id, maxheight)       # cam_vol is z-major
    if len(zs) >= 8:
        preps = list(_pool().map(prep1, list(enumerate(zs))))
    else:
        preps = [prep1(iz) for iz in enumerate(zs)]
    idx = [i for i, p in enumerate(preps) if p is not None]
    out_l, out_c = list(labels), list(cts)
    if not idx:
        return out_l, out_c, idx
    t = lambda key: torch.from_numpy(np.stack([preps[i][key] for i in idx])).unsqueeze(1).to(device)
    iv = lambda key: torch.tensor([preps[i][key] for i in idx], dtype=torch.int64, device=device)
    lab_t = torch.from_numpy(np.stack([labels[i] for i in idx]).astype(np.float32)).unsqueeze(1).to(device)
    ratio = torch.tensor([ratios[i] for i in idx], dtype=torch.float64, device=device)
    lab, ct, _ = synthesize(model, t('ct_masked'), t('mask'), t('cam'), ratio, t('ori_ct'), lab_t, iv('x1'), iv('x2'), iv('height'),
                            vert_id, maxheight)
    lab, ct = lab.cpu().numpy().astype(np.float64), ct.cpu().numpy().astype(np.float64)
    for j, i in enumerate(idx):
        out_l[i], out_c[i] = lab[j], ct[j]
    return out_l, out_c, idx


def process_volume(model, ct_data, label_data, cam_data, vert_id, device, maxheight=40):
    """process_nii_files' per-volume loop (reference :186-234) with the three chained syntheses (upper neighbour, lower
    neighbour, target) each batched over ALL z-slices: 3 generator launches per volume instead of ~130 bs=1 calls.
    ct_data in 0..255, label_data = vertebra ids, cam_data already scaled by 255 (reference :181).  Returns
    (output_ct [H,W,Z], output_seg [H,W,Z]) as float64 numpy arrays (zeros outside the processed z range)."""
    import numpy as np
    # z-major contiguous copies: every per-slice access below is then a contiguous 2-D array (the [H, W, Z] inputs have z fastest)
    lab_z = np.ascontiguousarray(np.moveaxis(label_data, 2, 0))
    ct_z = np.ascontiguousarray(np.moveaxis(ct_data, 2, 0))
    cam_z = np.ascontiguousarray(np.moveaxis(cam_data, 2, 0))
    zhas = np.flatnonzero((lab_z == vert_id).reshape(lab_z.shape[0], -1).any(axis=1))
    z0, z1 = int(zhas.min()), int(zhas.max())
    rng_len = z1 - z0 + 1
    new_len = int(rng_len * 4 / 5)
    nz0 = z0 + (rng_len - new_len) // 2
    nz1 = nz0 + new_len - 1
    centre = (nz0 + nz1) // 2
    zs = list(range(nz0, nz1 + 1))
    ratios = [abs(z - centre) / rng_len * 2 for z in zs]
    labels = [lab_z[z].copy() for z in zs]
    cts = [ct_z[z].copy() for z in zs]
    out_ct, out_seg = np.zeros(ct_z.shape, dtype=np.float64), np.zeros(ct_z.shape, dtype=np.float64)      # z-major, returned as [H, W, Z] views
    for nb, cond in ((vert_id - 1, vert_id > 8), (vert_id + 1, vert_id < 24)):
        cnt = (lab_z[nz0:nz1 + 1] == nb).reshape(len(zs), -1).sum(axis=1) if cond else None
        sel = [i for i in range(len(zs)) if cond and cnt[i] > 200]
        if sel:
            l2, c2, _ = _stage(model, cam_z, [labels[i] for i in sel], [cts[i] for i in sel], [zs[i] for i in sel],
                               [ratios[i] for i in sel], nb, device, maxheight)
            for j, i in enumerate(sel):
                labels[i], cts[i] = l2[j], c2[j]
    l3, c3, done = _stage(model, cam_z, labels, cts, zs, ratios, vert_id, device, maxheight)
    for i in done:
        out_seg[zs[i]], out_ct[zs[i]] = l3[i], c3[i]
    return np.moveaxis(out_ct, 0, 2), np.moveaxis(out_seg, 0, 2)
