"""Inference-side tensor work of the reference's eval_3d_sagittal_twostage.py (run_model :96-130, load_model :32-38)
on the HIP path: eval-mode generator forward, pred_h = ceil(pred2_h*maxheight), SHRM re-compositing of the CT
((x+1)*127.5) and of the label map (fine_seg > 0.5 -> vert_id), all on the device, batched over slices.

The reference runs ~130 sequential bs=1 forwards per volume, each with a PIL->tensor->H2D hop and a D2H copy
(SURVEY.md section 3.2); z-slices of one stage are independent, so `synthesize` takes a whole batch of slices.
`process_volume` keeps the slices on the device across the three chained stages, connected-component cleaning and band construction included.
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
    """Host mirror of the slice preparation in run_model (reference :46-98; the device path is hv_slice_components + hv_infer_prepare,
    pinned against this function and fixture G10): bbox of the vertebra, 41-row mask band (note `max_x+1`, :75), masked CT and
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


_PIN = {}
_POOL = []


def _pinned_f32(n, slot='buf'):
    b = _PIN.get(slot)
    if b is None or b.numel() < n:
        b = _PIN[slot] = torch.empty(n, dtype=torch.float32).pin_memory()
    return b[:n]


def _parallel(jobs):
    """Run a few numpy copy / convert jobs side by side (numpy releases the GIL inside them): the float64 <-> float32 slab copies of a
    volume are three / two independent passes over 3-13 MB each."""
    if not _POOL:
        from concurrent.futures import ThreadPoolExecutor
        _POOL.append(ThreadPoolExecutor(max_workers=3))
    for f in [_POOL[0].submit(j) for j in jobs]:
        f.result()


def _stage_device(model, st, vert_id, selected, maxheight):
    """One synthesis stage with the slices resident on the device (st: lab / ct / cam [S,H,W] fp32, ratio [S] fp64): component filter and
    row statistics (hv_slice_components), the generator's input planes (hv_infer_prepare), the batched generator + re-compositing, and
    the write-back of the slices that contained the vertebra (hv_select_slices).  Returns the valid flags ([S] int32, device); no host sync."""
    L = _lib.get()
    lab, ct, cam = st['lab'], st['ct'], st['cam']
    S, H, W = lab.shape
    dev = lab.device
    need = L.size('hv_slice_components_workspace_bytes', S, H, W)
    ws = st.get('ws')
    if ws is None or ws.numel() < need:
        ws = st['ws'] = torch.empty(need, dtype=torch.uint8, device=dev)
    stats = torch.empty(S, 4, dtype=torch.int32, device=dev)
    L.call('hv_slice_components', ptr(lab), S, H, W, ctypes.c_float(float(vert_id)), 50, ptr(stats), ptr(ws), ctypes.c_size_t(ws.numel()), stream())
    planes = torch.empty(4, S, 1, H, W, dtype=torch.float32, device=dev)
    x1, x2, height = (torch.empty(S, dtype=torch.int64, device=dev) for _ in range(3))
    valid = torch.empty(S, dtype=torch.int32, device=dev)
    L.call('hv_infer_prepare', ptr(ct), ptr(cam), ptr(stats), ptr(selected), S, H, W, int(maxheight), ptr(planes[0]), ptr(planes[1]), ptr(planes[2]),
           ptr(planes[3]), ptr(x1), ptr(x2), ptr(height), ptr(valid), stream())
    lab_o, ct_o, _ = synthesize(model, planes[0], planes[2], planes[3], st['ratio'], planes[1], lab.view(S, 1, H, W), x1, x2, height, vert_id, maxheight)
    per = ctypes.c_longlong(H * W)
    L.call('hv_select_slices', ptr(valid), ptr(lab_o), ptr(lab), S, per, 1, stream())
    L.call('hv_select_slices', ptr(valid), ptr(ct_o), ptr(ct), S, per, 1, stream())
    return valid


def process_volume(model, ct_data, label_data, cam_data, vert_id, device, maxheight=40):
    """process_nii_files' per-volume loop (reference :186-234) with the three chained syntheses (upper neighbour, lower
    neighbour, target) each batched over ALL z-slices: 3 generator launches per volume instead of ~130 bs=1 calls, and the slices
    stay on the device between the stages -- component filter, bounding rows, band re-stacking and quantisation (run_model :46-98) run there
    too (`_stage_device`).  ct_data in 0..255, label_data = vertebra ids, cam_data already scaled by 255 (reference :181).  Returns
    (output_ct [H,W,Z], output_seg [H,W,Z]) as float64 numpy arrays (zeros outside the processed z range).
    (Where the reference would raise -- a neighbour stage returning None, :212,:221 -- the slice passes through unchanged.)"""
    import numpy as np
    L = _lib.get()
    dev = torch.device(device)
    H, W, Z = label_data.shape
    has = [None] * 3                 # z-extent of the vertebra (:186-190): three row bands scanned side by side
    bands = np.array_split(np.arange(H), 3)

    def scan(i):
        has[i] = (label_data[bands[i][0]:bands[i][-1] + 1] == vert_id).any(axis=(0, 1)) if len(bands[i]) else np.zeros(Z, dtype=bool)
    _parallel([lambda i=i: scan(i) for i in range(3)])
    zhas = np.flatnonzero(has[0] | has[1] | has[2])
    z0, z1 = int(zhas.min()), int(zhas.max())
    rng_len = z1 - z0 + 1
    new_len = int(rng_len * 4 / 5)
    nz0 = z0 + (rng_len - new_len) // 2
    nz1 = nz0 + new_len - 1
    centre = (nz0 + nz1) // 2
    zs = list(range(nz0, nz1 + 1))
    S = len(zs)
    out_ct, out_seg = np.zeros((H, W, Z), dtype=np.float64), np.zeros((H, W, Z), dtype=np.float64)
    if S == 0:
        return out_ct, out_seg
    # the [H, W, Z] inputs have z fastest: the z-range is cut out as it lies (float32, [H*W][S]) and transposed to slices on the device
    stage = _pinned_f32(3 * H * W * S).view(3, H, W, S)          # one pinned staging buffer: no intermediate copies, asynchronous upload
    stage_np = stage.numpy()
    _parallel([(lambda i=i, vol=vol: np.copyto(stage_np[i], vol[:, :, nz0:nz1 + 1], casting='unsafe'))
               for i, vol in enumerate((label_data, ct_data, cam_data))])
    hws = stage.to(dev, non_blocking=True)
    vols = torch.empty(3, S, H, W, dtype=torch.float32, device=dev)
    L.call('hv_transpose_batched', ptr(hws), ptr(vols), 3, H * W, S, stream())
    st = {'lab': vols[0], 'ct': vols[1], 'cam': vols[2],
          'ratio': torch.tensor([abs(z - centre) / rng_len * 2 for z in zs], dtype=torch.float64, device=dev)}
    # the neighbour stages run on the slices where the neighbour has > 200 pixels on the ORIGINAL labels (reference :208,:217): both counts are
    # taken on the device before the first stage rewrites the label slices (hv_slice_count), one small read-back decides which stages run
    nbs = [nb for nb, cond in ((vert_id - 1, vert_id > 8), (vert_id + 1, vert_id < 24)) if cond]
    if nbs:
        counts = torch.empty(len(nbs), S, dtype=torch.int32, device=dev)
        for i, nb in enumerate(nbs):
            L.call('hv_slice_count', ptr(st['lab']), S, ctypes.c_longlong(H * W), ctypes.c_float(float(nb)), ptr(counts[i]), stream())
        sel = (counts > 200).int()
        runs = sel.any(dim=1).cpu()
        for i, nb in enumerate(nbs):
            if bool(runs[i]):
                _stage_device(model, st, nb, sel[i].contiguous(), maxheight)
    valid = _stage_device(model, st, vert_id, None, maxheight)
    res = torch.zeros(2, S, H * W, dtype=torch.float32, device=dev)
    per = ctypes.c_longlong(H * W)
    L.call('hv_select_slices', ptr(valid), ptr(st['ct']), ptr(res[0]), S, per, 0, stream())
    L.call('hv_select_slices', ptr(valid), ptr(st['lab']), ptr(res[1]), S, per, 0, stream())
    res_hws = torch.empty(2, H * W, S, dtype=torch.float32, device=dev)
    L.call('hv_transpose_batched', ptr(res), ptr(res_hws), 2, S, H * W, stream())
    back = _pinned_f32(2 * H * W * S, 'back').view(2, H * W, S)      # pinned: the download runs at PCIe speed, no staging copy
    back.copy_(res_hws, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    res_np = back.numpy().reshape(2, H, W, S)

    def put(dst, i):
        dst[:, :, nz0:nz1 + 1] = res_np[i]
    _parallel([lambda: put(out_ct, 0), lambda: put(out_seg, 1)])
    return out_ct, out_seg
