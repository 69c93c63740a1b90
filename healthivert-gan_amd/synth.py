"""Synthetic masked-vertebra slices with the batch-dict schema of the reference data loader.

Mimics the geometry produced by AlignedDataset.__getitem__ (reference
data/aligned_dataset.py:198-280) as specified in SURVEY.md section 8d: a smooth random CT
background with five stacked rounded-rectangle "vertebrae", the middle one being the target;
a fixed `h2`-row mask band centred on it; the masked CT re-stacked above/below the band; a
Gaussian "CAM" blob on a neighbour.  Pure numpy/scipy + torch tensors, no device work; shared by
the CPU oracle, the tests and bench.py so that all three see identical inputs for one seed.
"""
import numpy as np
import torch
from scipy.ndimage import gaussian_filter


def _rounded_rect(shape, r0, r1, c0, c1, rad=4):
    rr, cc = np.mgrid[0:shape[0], 0:shape[1]]
    dr = np.maximum(np.maximum(r0 + rad - rr, rr - (r1 - 1 - rad)), 0)
    dc = np.maximum(np.maximum(c0 + rad - cc, cc - (c1 - 1 - rad)), 0)
    inside = (rr >= r0) & (rr < r1) & (cc >= c0) & (cc < c1)
    return inside & (dr * dr + dc * dc <= rad * rad)


def make_slice(rng, size=256):
    """One sample as uint8/int fields, before ToTensor/Normalize."""
    s = size / 256.0
    h2 = int(round(40 * s))
    field = gaussian_filter(rng.standard_normal((size, size)), sigma=4 * s)
    field = field / (np.abs(field).max() + 1e-8) * 0.35
    ct = field.copy()
    pitch = int(round(28 * s))
    centre = size // 2 + int(rng.integers(-6, 7))
    ccol = size // 2 + int(rng.integers(-8, 9))
    label = np.zeros((size, size), np.int32)
    verts = []
    for k in range(-2, 3):
        hgt = int(round(rng.integers(20, 35) * s))
        if k == 0:
            hgt = min(hgt, h2 - 2)
        wid = int(round(rng.integers(36, 49) * s))
        r0 = centre + k * pitch - hgt // 2
        c0 = ccol - wid // 2
        m = _rounded_rect((size, size), r0, r0 + hgt, c0, c0 + wid, rad=max(2, int(4 * s)))
        ct[m] += 0.6
        label[m] = k + 3
        verts.append(m)
    ct = np.clip(ct, -1, 1)
    ct_u8 = np.round((ct + 1) * 127.5).astype(np.uint8)
    tgt = label == 3
    rows = np.where(tgt.any(axis=1))[0]
    x1, x2 = int(rows.min()), int(rows.max())
    height = x2 - x1
    mask_x = (x1 + x2) // 2
    if mask_x <= h2 // 2:
        min_x = 0
    elif size - mask_x <= h2 / 2:
        min_x = size - h2
    else:
        min_x = mask_x - h2 // 2
    max_x = min_x + h2

    def restack(img):
        out = np.zeros_like(img)
        out[:min_x] = img[(x1 - min_x):x1]
        out[max_x:] = img[x2:x2 + (size - max_x)]
        return out

    mask = np.zeros((size, size), np.uint8)
    mask[min_x:max_x] = 255
    normal = ((label > 0) & (label != 3)).astype(np.uint8) * 255
    nb = 2 if rng.random() < 0.5 else 4
    nr, nc = np.where(label == nb)
    rr, cc = np.mgrid[0:size, 0:size]
    cam = np.exp(-((rr - nr.mean()) ** 2 + (cc - nc.mean()) ** 2) / (2 * (12 * s) ** 2))
    cam_u8 = np.round(cam * 255).astype(np.uint8)
    return dict(A=ct_u8, B=restack(ct_u8), A_mask=tgt.astype(np.uint8) * 255, mask=mask,
                normal_vert=restack(normal), CAM=restack(cam_u8), height=height, x1=x1, x2=x2, h2=h2,
                slice_ratio=float(rng.uniform(0, 0.8)))


def make_batch(batch_size, size=256, seed=1234):
    """Batch dict exactly as default-collated from AlignedDataset:
    A,B f32 (B,1,S,S) in [-1,1]; A_mask,mask,normal_vert,CAM f32 in [0,1]; height,x1,x2,h2 int64 (B,);
    slice_ratio f64 (B,); A_paths/B_paths list[str]."""
    rng = np.random.default_rng(seed)
    items = [make_slice(rng, size) for _ in range(batch_size)]

    def img(key, norm):
        a = np.stack([it[key] for it in items]).astype(np.float32) / 255.0
        if norm:
            a = (a - 0.5) / 0.5
        return torch.from_numpy(a).unsqueeze(1)

    out = {'A': img('A', True), 'B': img('B', True), 'A_mask': img('A_mask', False), 'mask': img('mask', False),
           'normal_vert': img('normal_vert', False), 'CAM': img('CAM', False)}
    for k in ('height', 'x1', 'x2', 'h2'):
        out[k] = torch.tensor([it[k] for it in items], dtype=torch.int64)
    out['slice_ratio'] = torch.tensor([it['slice_ratio'] for it in items], dtype=torch.float64)
    out['A_paths'] = ['synthetic_%d_%d' % (seed, i) for i in range(batch_size)]
    out['B_paths'] = list(out['A_paths'])
    return out


def to_model_inputs(batch, direction='BtoA'):
    """The renaming Pix2PixModel.set_input performs (reference models/pix2pix_model.py:137-175)."""
    a_to_b = direction == 'AtoB'
    return dict(real_A=batch['A' if a_to_b else 'B'], real_B=batch['B' if a_to_b else 'A'],
                real_B_mask=batch['A_mask'], mask=batch['mask'], CAM=batch['CAM'], normal_vert=batch['normal_vert'],
                height=batch['height'], x1=batch['x1'], x2=batch['x2'], maxheight=batch['h2'],
                slice_ratio=batch['slice_ratio'])


def make_volume(nz=16, size=256, seed=7, target_id=20):
    """Synthetic straightened volume like the reference's datasets/straightened/{CT,label} + heat-map:
    returns (ct [size,size,nz] float in 0..255, label [size,size,nz] float vertebra ids target_id-2..target_id+2,
    cam [size,size,nz] float in 0..1).  The target vertebra spans the central z range, neighbours are present."""
    rng = np.random.default_rng(seed)
    base = make_slice(rng, size)
    field = gaussian_filter(rng.standard_normal((size, size, nz)), sigma=(4, 4, 1))
    field = field / (np.abs(field).max() + 1e-8) * 20
    ct = np.zeros((size, size, nz)); label = np.zeros((size, size, nz)); cam = np.zeros((size, size, nz))
    s = size / 256.0
    pitch, centre, ccol = int(round(28 * s)), size // 2, size // 2
    dims = [(int(round(rng.integers(22, 30) * s)), int(round(rng.integers(38, 46) * s))) for _ in range(5)]
    for z in range(nz):
        shrink = 1.0 - 0.25 * abs(z - (nz - 1) / 2) / (nz / 2)     # vertebra cross-section narrows towards the ends
        img = base['A'].astype(np.float64) * 0.6 + field[:, :, z] + 20
        lab = np.zeros((size, size))
        for k, (hgt, wid) in zip(range(-2, 3), dims):
            w2 = max(8, int(wid * shrink))
            r0, c0 = centre + k * pitch - hgt // 2, ccol - w2 // 2
            m = _rounded_rect((size, size), r0, r0 + hgt, c0, c0 + w2, rad=max(2, int(4 * s)))
            img[m] += 60
            lab[m] = target_id + k
        ct[:, :, z] = np.clip(img, 0, 255)
        label[:, :, z] = lab
        rr, cc = np.mgrid[0:size, 0:size]
        cam[:, :, z] = np.exp(-((rr - (centre - pitch)) ** 2 + (cc - ccol) ** 2) / (2 * (12 * s) ** 2))
    return ct, label, cam


def make_rhlv_pair(seed=0, H=64, W=64, Z=16, label_index=20, collapse=0.35, empty_ends=2, fake_shorter=False):
    """A (generated, original) pair of straightened label volumes [H, W, Z] like the inputs of the reference's
    evaluation/RHLV_quantification.py: the original vertebra (`label`) is wedge-compressed by `collapse` towards one side,
    the generated one (`fake`) has its restored height; voxels carry `label_index`, a neighbour vertebra carries
    label_index + 1, the first/last `empty_ends` slices are empty.  fake_shorter exercises the rescaling branch."""
    rng = np.random.default_rng(seed)
    fake = np.zeros((H, W, Z)); label = np.zeros((H, W, Z))
    c0, c1 = W // 4 + int(rng.integers(-3, 4)), 3 * W // 4 + int(rng.integers(-3, 4))
    r_mid = H // 2 + int(rng.integers(-2, 3))
    full_h = H // 3 + int(rng.integers(-2, 3))
    for z in range(empty_ends, Z - empty_ends):
        shrink = 1.0 - 0.3 * abs(z - (Z - 1) / 2) / (Z / 2)
        a, b = int(c0 + (1 - shrink) * 4), int(c1 - (1 - shrink) * 4)
        for w in range(a, b):
            t = (w - a) / max(1, b - a - 1)
            hf = int(round(full_h * shrink * (0.9 + 0.1 * np.sin(3 * t))))
            hl = int(round(hf * (1.0 - collapse * (1 - t)) + rng.integers(-1, 2)))
            if fake_shorter:
                hf, hl = max(1, hl - 2), hf
            fake[max(0, r_mid - hf // 2):r_mid + (hf + 1) // 2, w, z] = label_index
            label[max(0, r_mid - hl // 2):r_mid + (hl + 1) // 2, w, z] = label_index
        fake[2:6, a:b, z] = label_index + 1          # a neighbour: must be ignored by the == label_index test
        label[2:6, a:b, z] = label_index + 1
    return fake, label


def make_spine_volume(seed, H=96, W=56, Z=10, first_id=10, n_vert=5, pitch=18):
    """Synthetic straightened-spine volume for the batch-assembly path: ct in [0, 255) with fractional parts,
    label = vertebra ids stacked along the rows (extent varying with z, plus a few specks below the 50-pixel component filter),
    cam in [0, 1].  Returns float32 ct, uint8 label, float32 cam, all [H, W, Z] like the reference's NIfTI arrays."""
    import numpy as np
    rng = np.random.RandomState(seed)
    ct = (rng.rand(H, W, Z) * 255).astype(np.float32)
    cam = rng.rand(H, W, Z).astype(np.float32)
    label = np.zeros((H, W, Z), dtype=np.uint8)
    for v in range(n_vert):
        top = 3 + v * pitch
        for z in range(1, Z - 1):
            h = int(rng.randint(10, 15))
            w0 = int(rng.randint(4, 10))
            label[top + int(rng.randint(0, 3)):top + h, w0:W - w0, z] = first_id + v
        for _ in range(3):      # specks: fewer than 50 pixels, away from the body
            z, r, c = int(rng.randint(1, Z - 1)), top + 15, int(rng.randint(0, W - 4))
            label[r:r + 2, c:c + 3, z] = first_id + v
    return ct, label, cam
