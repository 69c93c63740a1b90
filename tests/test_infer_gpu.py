"""Batched inference tensor path (eval_3d_sagittal_twostage.run_model :96-130) against the CPU oracle."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_stage_batched_synthesis_matches_oracle():
    import hvgan
    from hvgan import synth, infer
    from hvgan.models.inpaint_networks import Generator
    from oracle import restate as R
    torch.manual_seed(3)
    net = Generator({'input_dim': 1, 'ngf': 16}, True)
    net.fine_generator.fc_height.bias.data.fill_(0.37)     # keep pred_h*40 away from an integer (the ceil() discontinuity)
    net.fine_generator.fc_height.weight.data.mul_(1e-2)
    net.cuda().train()
    b = synth.to_model_inputs(synth.make_batch(3, 256, seed=9))
    dev = torch.device('cuda:0')
    for _ in range(3):   # a never-trained spectral norm has random u/v (sigma far too small): let the power iteration settle
        net.run_forward(b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev), training=True)
    net.eval()
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    label = b['real_B_mask'] * 20.0
    lab, ct, pred = infer.synthesize(net, b['real_A'].to(dev), b['mask'].to(dev), b['CAM'].to(dev), b['slice_ratio'].to(dev),
                                     b['real_B'].to(dev), label.to(dev), b['x1'].to(dev), b['x2'].to(dev), b['height'].to(dev), 20)
    with torch.no_grad():
        (cs, fs, x1s, x2s, p1, p2), _ = R.generator_forward(sd, b['real_A'], b['mask'], 1 - b['CAM'], b['slice_ratio'], training=False)
    assert (pred.cpu() - p2.view(-1)).abs().max().item() <= 1e-3
    ph = p2.view(-1) * 40
    # keep the comparison away from the ceil() discontinuity
    assert all(abs(float(v) - round(float(v))) > 1e-3 for v in ph), ph
    ref_ct = (R.shrm_composite(x2s, b['real_B'], ph, b['height'], b['x1'], b['x2']) + 1) * 127.5
    assert (ct.cpu() - ref_ct[:, 0]).abs().max().item() <= 0.15          # 1e-3 in [-1,1] units
    seg = (fs > 0.5).float() * 20
    ref_lab = R.shrm_composite(seg, label, ph, b['height'], b['x1'], b['x2'])
    assert ((lab.cpu() - ref_lab[:, 0]).abs() > 0).float().mean().item() <= 1e-4
