"""Batch assembly with the volumes resident on the GPU.

Mirror of the reference's AlignedDataset.__getitem__ + default collate (data/aligned_dataset.py:100-146 slice choice, :176-280 per-item
arithmetic): the reference loads three float64 NIfTI volumes per item (~100 MB), cuts one slice, quantises it to uint8, re-stacks rows
around the masked band and converts to tensors on the CPU.  Here a vertebra's volumes are quantised ONCE (`VertebraVolume`, host, numpy),
uploaded once (`DeviceBatchAssembler`), and a batch is one `hv_assemble_batch` launch that writes the six float32 planes the model's
`set_input` consumes.  The 8-connected component filter the reference applies to every drawn slice (aligned_dataset.py:16-31,:186-188) runs
on the device too, once per volume over all of its slices (`hv_slice_components_u8`: filtered mask planes + pixel count / first row / last row
per slice -- everything the draw's acceptance test and the band rows need).  Only the slice draw itself stays on the host (np.random, same call
sequence as the reference => same slices for the same seed; it reads the per-slice table, no pixels).  `VertebraVolume.slice_info` without a
table is the host mirror of the filter (scipy.ndimage.label, the reference's own third-party call) that the tests hold the device table against.
"""
import ctypes

import numpy as np
import torch

from . import lib as _lib
from .lib import ptr, stream


def _remove_small_components(a, min_size):
    """aligned_dataset.py:16-31."""
    from scipy.ndimage import label
    labeled, n = label(a, np.ones((3, 3), dtype=np.int32))
    for i in range(1, n + 1):
        if np.sum(labeled == i) < min_size:
            a[labeled == i] = 0
    return a


def band_rows(x1, x2, width, h2):
    """aligned_dataset.py:214-226 -> (min_x, max_x)."""
    mask_x = (x1 + x2) // 2
    if mask_x <= h2 // 2:
        return 0, h2
    if width - mask_x <= h2 / 2:
        return width - h2, width
    return mask_x - h2 // 2, mask_x - h2 // 2 + h2


class VertebraVolume:
    """Host side of one vertebra volume: the uint8 planes the reference derives per item, z-major ([Z][H][W], a slice is contiguous).

    ct_data, label_data, cam_data: [H, W, Z] arrays as `nib.load(...).get_fdata()` returns them (cam unscaled; the reference multiplies
    it by 255 at :167).  normal_vert_list: ids (str or int) of the patient's normal vertebrae (:178, :190-196)."""

    def __init__(self, ct_data, label_data, cam_data, vert_id, normal_vert_list, path='', maxheight=40, view='sagittal'):
        # view: which in-plane axis the slices run along.  'sagittal' = the reference loader's `[:, :, z]` (data/aligned_dataset.py:149-200);
        # 'coronal' = the axis-swapped slicing `[:, z, :]` of the reference's coronal scripts (evaluation/RHLV_quantification_coronal.py:51-54,
        # generation_eval_coronal.py) -- BASELINE config #5 feeds the same networks slices of both views
        if view not in ('sagittal', 'coronal'):
            raise ValueError("view must be 'sagittal' or 'coronal'")
        self.view = view
        if view == 'coronal':
            ct_data, label_data, cam_data = (np.swapaxes(np.asarray(v), 1, 2) for v in (ct_data, label_data, cam_data))
        label_data = np.asarray(label_data, dtype=np.float64)
        self.path, self.vert_id, self.maxheight = path, int(vert_id), maxheight
        self.H, self.W, self.Z = label_data.shape
        vert = np.zeros_like(label_data)
        vert[label_data == self.vert_id] = 1
        normal = label_data.copy()
        if normal_vert_list:
            for nv in normal_vert_list:
                normal[normal == int(nv)] = 255
            normal[normal != 255] = 0
        else:
            normal = np.zeros_like(label_data)
        zmaj = lambda v: np.ascontiguousarray(np.moveaxis(v, 2, 0))
        self.ct = zmaj(np.asarray(ct_data, dtype=np.float64).astype(np.uint8))
        self.cam = zmaj((np.asarray(cam_data, dtype=np.float64) * 255).astype(np.uint8))
        self.normal = zmaj(normal.astype(np.uint8))
        self._vert = zmaj(vert)                                  # float64 0/1, filtered slice by slice as slices are drawn
        self.vert = (self._vert * 255).astype(np.uint8)
        zs = np.flatnonzero(self._vert.reshape(self.Z, -1).any(axis=1))
        if zs.size == 0:
            raise ValueError("vertebra %d is absent from the label volume" % self.vert_id)
        self.z0, self.z1 = int(zs.min()), int(zs.max())
        self._info = {}                                          # z -> (pixel count, x1, x2) of the filtered slice
        self.dirty = set()                                       # slices whose filtered mask differs from what was uploaded
        self._choice = None

    def set_slice_table(self, stats):
        """stats: [Z][4] ints from hv_slice_components_u8 (count, first row, last row, row sum) -> the per-slice table the draw reads."""
        for z in range(self.Z):
            c, r0, r1 = int(stats[z][0]), int(stats[z][1]), int(stats[z][2])
            self._info[z] = (float(c), r0 if c else -1, r1 if c else -1)
        self.dirty.clear()

    def slice_info(self, z):
        info = self._info.get(z)
        if info is None:      # host mirror of the device filter (stand-alone use / tests)
            before = self._vert[z].copy()
            _remove_small_components(self._vert[z], 50)
            if not np.array_equal(before, self._vert[z]):
                self.vert[z] = (self._vert[z] * 255).astype(np.uint8)
                self.dirty.add(z)
            rows = np.flatnonzero(self._vert[z].any(axis=1))
            info = (float(self._vert[z].sum()), int(rows.min()) if rows.size else -1, int(rows.max()) if rows.size else -1)
            self._info[z] = info
        return info

    def prefilter(self):
        """Run the component filter on every slice the weighted draw can return (the central 4/5 of the z-extent)."""
        self.weighted_random_slice_range()
        for z in self._choice[0]:
            self.slice_info(z)

    def weighted_random_slice_range(self):
        """(candidate slices, their probabilities, centre, z-extent) of aligned_dataset.py:100-121; depends on (z0, z1) only."""
        if self._choice is None:
            z0, z1 = self.z0, self.z1
            range_length = z1 - z0 + 1
            new_range_length = int(range_length * 4 / 5)
            new_z0 = z0 + (range_length - new_range_length) // 2
            new_z1 = new_z0 + new_range_length - 1
            center = (new_z0 + new_z1) // 2
            weights = [1 - abs(i - center) / (new_z1 - new_z0) for i in range(new_z0, new_z1 + 1)]
            total = sum(weights)
            self._choice = (range(new_z0, new_z1 + 1), [w / total for w in weights], center, range_length)
        return self._choice

    def weighted_random_slice(self):
        """aligned_dataset.py:100-124 (one np.random.choice per call)."""
        rng, p, center, range_length = self.weighted_random_slice_range()
        idx = int(np.random.choice(rng, p=p))
        return idx, abs(idx - center) / range_length * 2

    def draw(self):
        """aligned_dataset.py:126-146 -> (z, slice_ratio, x1, x2)."""
        for _ in range(100):
            z, ratio = self.weighted_random_slice()
            count, x1, x2 = self.slice_info(z)
            if count > 50 and x2 - x1 < self.maxheight:
                return z, ratio, x1, x2
        raise ValueError("Failed to find a non-empty slice after 100 attempts.")


class DeviceBatchAssembler:
    """The data set on the GPU: `batch(indices)` returns the dict the reference's DataLoader would collate from
    `AlignedDataset.__getitem__(i) for i in indices`, image tensors already on the device (`Pix2PixModel.set_input` takes it as is)."""

    IMAGE_KEYS = ('A', 'B', 'A_mask', 'mask', 'normal_vert', 'CAM')

    def __init__(self, volumes, device):
        self.volumes = list(volumes)
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise _lib.HipLibraryMissing("batch assembly runs on the GPU (libhvgan.so); there is no CPU fallback")
        shapes = {(v.H, v.W) for v in self.volumes}
        if len(shapes) != 1:
            raise ValueError("all volumes of one assembler must share the slice size, got %s" % sorted(shapes))
        (self.H, self.W), = shapes
        self._planes = []         # per volume: uint8 tensor [4][Z][H][W] = ct, vert, normal, cam
        self._L = L = _lib.get()
        for v in self.volumes:
            planes = torch.from_numpy(np.stack([v.ct, v.vert, v.normal, v.cam])).to(self.device)
            # component filter of EVERY slice on the device, in place on the resident mask plane, before the first draw: the training loop never
            # touches pixels on the host and never refreshes a plane
            need = L.size('hv_slice_components_workspace_bytes', v.Z, v.H, v.W)
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            stats = torch.empty(v.Z, 4, dtype=torch.int32, device=self.device)
            L.call('hv_slice_components_u8', ptr(planes[1]), v.Z, v.H, v.W, 255, 50, ptr(stats), ptr(planes[1]), ptr(ws),
                   ctypes.c_size_t(need), stream())
            v.set_slice_table(stats.cpu().numpy())
            self._planes.append(planes)

    def __len__(self):
        return len(self.volumes)

    def _pinned(self, nbytes):
        """Next buffer of a small ring of pinned staging buffers (a buffer is reused only four batches later, long after its copy ran)."""
        self._ring_i = (getattr(self, '_ring_i', -1) + 1) % 4
        ring = self.__dict__.setdefault('_ring', [None] * 4)
        if ring[self._ring_i] is None or ring[self._ring_i].numel() < nbytes:
            ring[self._ring_i] = torch.empty(max(nbytes, 4096), dtype=torch.uint8).pin_memory()
        return ring[self._ring_i]

    def batch(self, indices):
        L = self._L
        B, H, W = len(indices), self.H, self.W
        items = (L.hv_assemble_item * B)()
        meta = {'height': [], 'x1': [], 'x2': [], 'h2': [], 'slice_ratio': [], 'slice': []}
        for b, i in enumerate(indices):
            v = self.volumes[i]
            z, ratio, x1, x2 = v.draw()
            min_x, max_x = band_rows(x1, x2, H, v.maxheight)
            if x1 - min_x < 0 or x2 > max_x or max_x > H or min_x < 0:
                raise ValueError("could not broadcast: vertebra rows [%d, %d] do not fit the band [%d, %d) of a %d-row slice"
                                 % (x1, x2, min_x, max_x, H))           # the reference fails with numpy's broadcasting ValueError here
            base = self._planes[i].data_ptr()
            plane = self._planes[i].shape[1] * H * W
            it = items[b]
            it.ct, it.vert, it.normal, it.cam = (base + k * plane + z * H * W for k in range(4))
            it.x1, it.x2, it.min_x, it.max_x = x1, x2, min_x, max_x
            for k, val in (('height', x2 - x1), ('x1', x1), ('x2', x2), ('h2', v.maxheight), ('slice_ratio', ratio), ('slice', z)):
                meta[k].append(val)
        # one packed, pinned upload per batch (descriptor table + integer / float64 metadata): a pageable H2D copy would block the host until the
        # previous step's kernels have drained, and the host-side slice draws of this batch would no longer overlap them
        isz = ctypes.sizeof(L.hv_assemble_item) * B
        total = isz + 5 * 8 * B
        pin = self._pinned(total)
        pin[:isz] = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        pin[isz:isz + 32 * B].view(torch.int64).copy_(torch.tensor([meta[k] for k in ('height', 'x1', 'x2', 'h2')], dtype=torch.int64).reshape(-1))
        pin[isz + 32 * B:total].view(torch.float64).copy_(torch.tensor(meta['slice_ratio'], dtype=torch.float64))
        dev = torch.empty(total, dtype=torch.uint8, device=self.device)
        dev.copy_(pin[:total], non_blocking=True)
        d_items = dev[:isz]
        out = torch.empty(6, B, 1, H, W, dtype=torch.float32, device=self.device)
        L.call('hv_assemble_batch', ctypes.cast(ptr(d_items), ctypes.POINTER(L.hv_assemble_item)), B, H, W,
               *(ptr(out[k]) for k in range(6)), stream())
        batch = {k: out[n] for n, k in enumerate(self.IMAGE_KEYS)}
        ints = dev[isz:isz + 32 * B].view(torch.int64).view(4, B)
        for n, k in enumerate(('height', 'x1', 'x2', 'h2')):
            batch[k] = ints[n]
        batch['slice_ratio'] = dev[isz + 32 * B:total].view(torch.float64)
        batch['slice'] = meta['slice']
        batch['A_paths'] = [self.volumes[i].path for i in indices]
        batch['B_paths'] = list(batch['A_paths'])
        self._keep = dev         # the descriptor table must outlive the asynchronous launch
        return batch
