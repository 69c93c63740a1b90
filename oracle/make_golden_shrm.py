"""Fixture G12 ("SHRM compositing on hand-built inputs incl. pred_h just above/below an integer"): the REFERENCE's own
Pix2PixModel.forward (models/pix2pix_model.py:180-264) run on hand-built generator outputs.

    python oracle/make_golden_shrm.py [--ref /root/reference] [--out tests/golden]

The model is the reference's Pix2PixModel; only its generator is replaced by a stand-in that returns prescribed tensors, so everything behind
netG -- pred_h = pred * maxheight, the two thresholds, both SHRM compositing loops with math.ceil / max(height) / height_diff // 2, the local
crops and both Sobel edge maps -- is the reference's code on inputs chosen at the decision boundaries:
  sample 0  pred2 * 40 just BELOW an integer (29.99999 -> 30)      sample 4  band at the top edge (x_upper == 0)
  sample 1  pred2 * 40 just ABOVE an integer (30.00001 -> 31)      sample 5  band at the bottom edge (x2 close to 256)
  sample 2  pred_h < height (the measured height wins)             sample 6  odd height_diff (x_upper uses height_diff // 2)
  sample 3  pred * 40 an exact integer                             sample 7  pred == 1 (the maximum height, 40)
The stage-1 image uses the same cases in another order.  Nothing of the reference is copied: the fixture holds inputs and outputs.
TEST INFRASTRUCTURE ONLY."""
import argparse
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def build_inputs():
    B, H, W = 8, 256, 256
    r = torch.arange(H, dtype=torch.float32).view(1, 1, H, 1)
    c = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W)
    n = torch.arange(B, dtype=torch.float32).view(B, 1, 1, 1)
    pat = lambda a, b_, m: ((r * a + c * b_ + n * 17) % m) / (m / 2.0) - 1.0          # deterministic, every row / column / sample different
    real_B = pat(7, 3, 251)
    x_stage1 = pat(5, 11, 241)
    x_stage2 = pat(13, 2, 239)
    height = torch.tensor([24, 24, 26, 28, 22, 25, 23, 30], dtype=torch.int64)
    x1 = torch.tensor([100, 90, 110, 80, 2, 226, 120, 60], dtype=torch.int64)
    x2 = x1 + height
    maxh = torch.full((B,), 40, dtype=torch.int64)
    p2 = torch.tensor([29.99999, 30.00001, 20.3, 28.0, 26.0, 29.5, 29.2, 40.0], dtype=torch.float64) / 40.0
    p1 = torch.tensor([30.00001, 29.99999, 28.0, 20.3, 25.7, 27.0, 26.0, 39.99999], dtype=torch.float64) / 40.0
    pred2 = p2.float().view(B, 1)
    pred1 = p1.float().view(B, 1)
    # sigmoid maps around the 0.5 threshold (exact 0.5 stays 0 in the reference: '>')
    fine = ((r * 3 + c * 5 + n) % 7) / 6.0
    coarse = ((r * 2 + c * 7 + n * 3) % 5) / 4.0
    mask = torch.zeros(B, 1, H, W)
    for i in range(B):
        mx = int((x1[i] + x2[i]) // 2)
        mask[i, :, max(0, mx - 20):min(H, mx + 20)] = 1
    cam = pat(1, 1, 97) * 0.5 + 0.5
    real_B_mask = (((r + c + n) % 9) < 4).float().expand(B, 1, H, W).contiguous()
    normal_vert = (((r * 2 + c + n) % 11) < 5).float().expand(B, 1, H, W).contiguous()
    return dict(real_B=real_B.expand(B, 1, H, W).contiguous(), x_stage1=x_stage1.expand(B, 1, H, W).contiguous(),
                x_stage2=x_stage2.expand(B, 1, H, W).contiguous(), fine=fine.expand(B, 1, H, W).contiguous(),
                coarse=coarse.expand(B, 1, H, W).contiguous(), mask=mask, cam=cam.expand(B, 1, H, W).contiguous(), real_B_mask=real_B_mask,
                normal_vert=normal_vert, height=height, x1=x1, x2=x2, maxheight=maxh, pred1=pred1, pred2=pred2)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(ROOT, 'tests', 'golden'))
    args = ap.parse_args()
    inp, nets, edge, unet, p2p = MG.import_reference(args.ref)
    torch.manual_seed(5)
    model = p2p.Pix2PixModel(MG.make_opt())
    I = build_inputs()
    B = I['real_B'].shape[0]

    class StandIn(torch.nn.Module):          # returns the prescribed 7-tuple (reference models/inpaint_networks.py:28-32)
        def forward(self, x, mask, cam, ratio):
            return I['coarse'], I['fine'], I['x_stage1'], I['x_stage2'], torch.zeros(B, 3, 256, 256), I['pred1'], I['pred2']

    model.netG = StandIn()
    batch = {'A': I['real_B'], 'B': I['real_B'] * (1 - I['mask']) - I['mask'], 'A_mask': I['real_B_mask'], 'mask': I['mask'], 'CAM': I['cam'],
             'normal_vert': I['normal_vert'], 'height': I['height'], 'x1': I['x1'], 'x2': I['x2'], 'h2': I['maxheight'],
             'slice_ratio': torch.linspace(0.1, 0.8, B, dtype=torch.float64), 'A_paths': [''] * B, 'B_paths': [''] * B}
    model.set_input(batch)
    with torch.no_grad():
        model.forward()
    keep = lambda t: t.detach()[..., ::4].contiguous()          # every 4th column: the compositing acts on whole rows
    out = {k: keep(getattr(model, k)) for k in ('fake_B', 'fake_B_coarse', 'fake_B_local', 'real_B_local', 'fake_B_mask_raw', 'coarse_seg_binary',
                                                'real_edges', 'fake_edges')}
    out['pred1_h'] = model.pred1_h.detach()
    out['pred2_h'] = model.pred2_h.detach()
    ins = {k: (v if v.dim() < 4 else keep(v)) for k, v in I.items()}
    MG.save(args.out, 'g12_shrm', cols=np.int64(4), **{'in': ins, 'res': out})
    print('wrote g12_shrm.npz', {k: tuple(v.shape) for k, v in out.items()})


if __name__ == '__main__':
    main()
