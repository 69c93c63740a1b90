"""Helpers with the reference's names (reference models/inpaint_tools.py:7-70, :73-100).  The HIP path does not
use them (patch extraction lives in hv_ca_patches / hv_ca_raw_patches); they are thin tensor-shape utilities kept
for callers that import them, plus the flow colour coding used only for visualisation."""
import numpy as np
import torch
import torch.nn.functional as F


def same_padding(images, ksizes, strides, rates):
    assert len(images.size()) == 4
    _, _, rows, cols = images.size()
    pads = []
    for n, k, s, r in ((rows, ksizes[0], strides[0], rates[0]), (cols, ksizes[1], strides[1], rates[1])):
        out = (n + s - 1) // s
        total = max(0, (out - 1) * s + (k - 1) * r + 1 - n)
        pads.append((total // 2, total - total // 2))
    return F.pad(images, (pads[1][0], pads[1][1], pads[0][0], pads[0][1]))


def extract_image_patches(images, ksizes, strides, rates, padding='same'):
    assert len(images.size()) == 4
    assert padding in ['same', 'valid']
    if padding == 'same':
        images = same_padding(images, ksizes, strides, rates)
    return F.unfold(images, kernel_size=ksizes, dilation=rates, padding=0, stride=strides)


def reduce_mean(x, axis=None, keepdim=False):
    for i in sorted(axis or range(x.dim()), reverse=True):
        x = torch.mean(x, dim=i, keepdim=keepdim)
    return x


def reduce_sum(x, axis=None, keepdim=False):
    for i in sorted(axis or range(x.dim()), reverse=True):
        x = torch.sum(x, dim=i, keepdim=keepdim)
    return x


def _color_wheel():
    segs = ((15, 0, 1, 1), (6, 1, 0, -1), (4, 1, 2, 1), (11, 2, 1, -1), (13, 2, 0, 1), (6, 0, 2, -1))
    wheel = np.zeros((sum(s[0] for s in segs), 3))
    row = 0
    for n, full, ramp, sign in segs:
        t = np.floor(255 * np.arange(n) / n)
        wheel[row:row + n, full] = 255
        wheel[row:row + n, ramp] = t if sign > 0 else 255 - t
        row += n
    return wheel


def flow_to_image(flow):
    """(B,h,w,2) integer offsets -> (B,h,w,3) float32 Middlebury colour coding with the running max radius of the
    reference (visualisation only; CPU numpy)."""
    wheel, out, maxrad = _color_wheel(), [], -1.0
    ncols = wheel.shape[0]
    for i in range(flow.shape[0]):
        u, v = flow[i, :, :, 0].astype(np.float64), flow[i, :, :, 1].astype(np.float64)
        bad = (np.abs(u) > 1e7) | (np.abs(v) > 1e7)
        u[bad] = 0
        v[bad] = 0
        maxrad = max(maxrad, float(np.sqrt(u ** 2 + v ** 2).max()))
        u, v = u / (maxrad + np.finfo(float).eps), v / (maxrad + np.finfo(float).eps)
        rad = np.sqrt(u ** 2 + v ** 2)
        fk = (np.arctan2(-v, -u) / np.pi + 1) / 2 * (ncols - 1) + 1
        k0 = np.floor(fk).astype(int)
        k1 = np.where(k0 + 1 == ncols + 1, 1, k0 + 1)
        f = fk - k0
        img = np.zeros(u.shape + (3,))
        for c in range(3):
            col = (1 - f) * wheel[k0 - 1, c] / 255 + f * wheel[k1 - 1, c] / 255
            inside = rad <= 1
            col = np.where(inside, 1 - rad * (1 - col), col * 0.75)
            img[:, :, c] = np.uint8(np.floor(255 * col))
        out.append(img)
    return np.float32(np.uint8(out))
