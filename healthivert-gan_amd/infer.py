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


_POOL = []


def _parallel(jobs):
    """Run a few numpy copy jobs side by side (numpy releases the GIL inside them): the pinned copies of a volume's three float64 arrays."""
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


class VolumePipeline:
    """process_nii_files' per-volume loop (reference eval_3d_sagittal_twostage.py:186-241) for a STREAM of volumes.

    Per volume the host does ONE thing with the data: it copies the three float64 [H, W, Z] arrays into a pinned buffer (three threads; numpy
    releases the GIL).  Everything else runs on the device: the upload (copy stream), the z-extent scan and the neighbours' pixel counts
    (hv_volume_scan, one pass over the label volume), the z-range cut + float32 conversion + transpose to slices (hv_volume_slices), the three
    chained synthesis stages batched over all slices (`_stage_device`), the float64 output volumes (hv_volume_merge) and their download.  Buffers
    are double: while the compute stream runs volume n's stages, volume n + 1 is copied to pinned memory (worker thread), uploaded and scanned
    (copy stream) and volume n - 1's outputs come down (output stream) -- one host wait per volume (its 3 x Z scan counts decide S, the z-range and
    which neighbour stages run)."""

    def __init__(self, model, device, maxheight=40):
        from concurrent.futures import ThreadPoolExecutor
        from . import engine
        self.model, self.dev, self.maxheight = model, torch.device(device), maxheight
        self.copy_stream = engine.named_stream('volume-upload', self.dev)
        self.out_stream = engine.named_stream('volume-download', self.dev)
        self.slots = [None, None]
        self.stager = ThreadPoolExecutor(max_workers=1)

    def _slot(self, i, H, W, Z):
        sl = self.slots[i]
        if sl is None or sl['shape'] != (H, W, Z):
            n = H * W * Z
            sl = self.slots[i] = {
                'shape': (H, W, Z),
                'pin_in': torch.empty(3 * n, dtype=torch.float64).pin_memory(), 'dev_in': torch.empty(3 * n, dtype=torch.float64, device=self.dev),
                'counts': torch.empty(3 * Z, dtype=torch.int32, device=self.dev), 'pin_counts': torch.empty(3 * Z, dtype=torch.int32).pin_memory(),
                'dev_out': torch.empty(2 * n, dtype=torch.float64, device=self.dev), 'pin_out': torch.empty(2 * n, dtype=torch.float64).pin_memory(),
                'ev_in': torch.cuda.Event(), 'ev_done': torch.cuda.Event(), 'ev_out': torch.cuda.Event(), 'ev_free': None}
        return sl

    def _stage_in(self, i, ct_data, label_data, cam_data, vert_id):
        """Worker thread: the volume's three arrays -> pinned slot i -> device; scan; counts -> pinned.  Returns the slot (its ev_in says when)."""
        import numpy as np
        H, W, Z = label_data.shape
        sl = self._slot(i, H, W, Z)
        if sl['ev_free'] is not None:
            sl['ev_free'].synchronize()          # the compute stream has read this slot's previous volume out of dev_in
        n = H * W * Z
        pin = sl['pin_in'].numpy().reshape(3, H, W, Z)
        _parallel([(lambda k=k, vol=vol: np.copyto(pin[k], vol, casting='unsafe')) for k, vol in enumerate((label_data, ct_data, cam_data))])
        L = _lib.get()
        nbs = [float(nb) if cond else -1.0 for nb, cond in ((vert_id - 1, vert_id > 8), (vert_id + 1, vert_id < 24))]
        with torch.cuda.stream(self.copy_stream):
            sl['dev_in'].copy_(sl['pin_in'], non_blocking=True)
            L.call('hv_volume_scan', ptr(sl['dev_in']), ctypes.c_longlong(H * W), Z, ctypes.c_double(float(vert_id)), ctypes.c_double(nbs[0]),
                   ctypes.c_double(nbs[1]), ptr(sl['counts']), stream())
            sl['pin_counts'].copy_(sl['counts'], non_blocking=True)
            sl['ev_in'].record(self.copy_stream)
        sl['vert_id'] = vert_id
        return sl

    def _compute(self, sl):
        """Main thread: wait for the slot's scan counts, plan the z-range, queue the stages, the output volumes and their download."""
        L = _lib.get()
        H, W, Z = sl['shape']
        vert_id, dev = sl['vert_id'], self.dev
        sl['ev_in'].synchronize()                # the only host wait of the volume (the upload + scan ran while the previous volume computed)
        counts = sl['pin_counts'].numpy().reshape(3, Z)
        zhas = counts[0].nonzero()[0]
        main = torch.cuda.current_stream(dev)
        n = H * W * Z
        out = sl['dev_out'].view(2, n)
        sl['empty'] = zhas.size == 0
        S = 0
        if not sl['empty']:
            z0, z1 = int(zhas.min()), int(zhas.max())
            rng_len = z1 - z0 + 1
            new_len = int(rng_len * 4 / 5)
            nz0 = z0 + (rng_len - new_len) // 2
            nz1 = nz0 + new_len - 1
            centre = (nz0 + nz1) // 2
            S = nz1 - nz0 + 1
        if S <= 0:
            sl['empty'] = True
            sl['dev_out'].zero_()
        else:
            main.wait_event(sl['ev_in'])
            vols = torch.empty(3, S, H, W, dtype=torch.float32, device=dev)
            for k in range(3):
                L.call('hv_volume_slices', ptr(sl['dev_in'][k * n:(k + 1) * n]), ctypes.c_longlong(H * W), Z, nz0, S, ptr(vols[k]), stream())
            sl['ev_free'] = torch.cuda.Event()
            sl['ev_free'].record(main)           # dev_in may take the next volume
            st = {'lab': vols[0], 'ct': vols[1], 'cam': vols[2],
                  'ratio': torch.tensor([abs(z - centre) / rng_len * 2 for z in range(nz0, nz1 + 1)], dtype=torch.float64, device=dev)}
            # the neighbour stages run on the slices where the neighbour has > 200 pixels on the ORIGINAL labels (reference :208,:217): the scan counted
            # them per z, so the host already knows which stages run and on which slices
            for j, (nb, cond) in enumerate(((vert_id - 1, vert_id > 8), (vert_id + 1, vert_id < 24))):
                if cond:
                    sel = counts[1 + j, nz0:nz1 + 1] > 200
                    if sel.any():
                        _stage_device(self.model, st, nb, torch.from_numpy(sel.astype('int32')).to(dev, non_blocking=True), self.maxheight)
            valid = _stage_device(self.model, st, vert_id, None, self.maxheight)
            for k, name in enumerate(('ct', 'lab')):
                L.call('hv_volume_merge', ptr(st[name]), ptr(valid), ctypes.c_longlong(H * W), Z, nz0, S, ptr(out[k]), stream())
        sl['ev_done'].record(main)
        with torch.cuda.stream(self.out_stream):
            self.out_stream.wait_event(sl['ev_done'])
            sl['pin_out'].copy_(sl['dev_out'], non_blocking=True)
            sl['ev_out'].record(self.out_stream)

    def _finish(self, sl, copy):
        import numpy as np
        H, W, Z = sl['shape']
        sl['ev_out'].synchronize()
        res = sl['pin_out'].numpy().reshape(2, H, W, Z)
        if not copy:
            return res[0], res[1]
        outs = [np.empty((H, W, Z), dtype=np.float64), np.empty((H, W, Z), dtype=np.float64)]
        _parallel([lambda: np.copyto(outs[0], res[0]), lambda: np.copyto(outs[1], res[1])])
        return outs[0], outs[1]

    def run(self, volumes, copy=True):
        """volumes: iterable of (ct_data, label_data, cam_data, vert_id) -- float64 [H, W, Z] arrays, ct in 0..255, cam already scaled by 255
        (reference :181).  Yields (output_ct, output_seg) [H, W, Z] float64 per volume, in order.  copy=False hands out views of the pinned download
        buffers instead of fresh arrays: a view is valid only until the generator is advanced again -- the NEXT `next()` call queues the download of the
        volume after next into the same pinned slot (two slots), before the following result is yielded.  Consume (or copy) each result before asking
        for the next one."""
        it = iter(volumes)
        nxt = next(it, None)
        if nxt is None:
            return
        fut = self.stager.submit(self._stage_in, 0, *nxt)
        i, prev = 0, None
        while fut is not None:
            sl = fut.result()
            nxt = next(it, None)
            fut = self.stager.submit(self._stage_in, (i + 1) & 1, *nxt) if nxt is not None else None
            self._compute(sl)
            if prev is not None:
                yield self._finish(prev, copy)
            prev, i = sl, i + 1
        yield self._finish(prev, copy)


_PIPELINES = {}


def process_volumes(model, volumes, device, maxheight=40, copy=True):
    """Generator over (output_ct, output_seg) for an iterable of (ct_data, label_data, cam_data, vert_id): `VolumePipeline.run`."""
    key = (id(model), str(device), maxheight)
    pipe = _PIPELINES.get(key)
    if pipe is None:
        pipe = _PIPELINES[key] = VolumePipeline(model, device, maxheight)
    return pipe.run(volumes, copy=copy)


def process_volume(model, ct_data, label_data, cam_data, vert_id, device, maxheight=40):
    """process_nii_files' per-volume body (reference :186-234) with the three chained syntheses (upper neighbour, lower neighbour, target) each
    batched over ALL z-slices: 3 generator launches per volume instead of ~130 bs=1 calls, the slices staying on the device between the stages --
    component filter, bounding rows, band re-stacking and quantisation (run_model :46-98) included.  ct_data in 0..255, label_data = vertebra
    ids, cam_data already scaled by 255 (reference :181).  Returns (output_ct [H,W,Z], output_seg [H,W,Z]) as float64 numpy arrays (zeros outside
    the processed z range).  One volume through `VolumePipeline`; a loop over volumes should use `process_volumes`, which overlaps their transfers.
    (Where the reference would raise -- a neighbour stage returning None, :212,:221 -- the slice passes through unchanged.)"""
    return next(process_volumes(model, [(ct_data, label_data, cam_data, vert_id)], device, maxheight))
