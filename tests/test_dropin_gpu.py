"""Drop-in boundary: with healthivert-gan_amd/ first on PYTHONPATH, `import models` (the reference's own package name)
resolves to the HIP-backed package, and the calls train.py / eval_3d_sagittal_twostage.py make work unchanged."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import torch
from argparse import Namespace
import models                                           # reference: `from models import create_model` (train.py:30)
from models.inpaint_networks import Generator            # reference: eval_3d_sagittal_twostage.py:11
assert 'healthivert-gan_amd' in models.__file__
import sys; sys.path.insert(0, %r)
from hvgan import synth
opt = Namespace(model='pix2pix', gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/hv_dropin', name='t', preprocess='none', input_nc=1,
                output_nc=1, ngf=64, ndf=64, netD='basic', netG='unet_256', n_layers_D=3, norm='batch', init_type='normal', init_gain=0.02,
                no_dropout=True, gan_mode='vanilla', lr=2e-4, beta1=0.5, lambda_L1=200.0, direction='BtoA', lr_policy='linear',
                epoch_count=1, n_epochs=100, n_epochs_decay=100, continue_train=False, load_iter=0, epoch='latest', verbose=False)
model = models.create_model(opt)
model.setup(opt)
model.update_learning_rate()
model.set_input(synth.make_batch(2, 256, seed=3))
model.optimize_parameters()
losses = model.get_current_losses()
assert set(losses) == {'G_GAN','G_maskL1','G_Dice','coarse_Dice','edge','D_real_1','D_fake_1','D_real_2','D_fake_2','D_real_3','D_fake_3','h'}
assert all(v == v for v in losses.values())
vis = model.get_current_visuals()
assert len(vis) == 13 and all(t.shape == (2, 1, 256, 256) for t in vis.values())
model.save_networks('latest')
# evaluate_model-style use (train.py:56-99): eval mode, direct netG call, 7-tuple
model.eval()
with torch.no_grad():
    out = model.netG(model.real_A, model.mask, 1 - model.CAM, model.slice_ratio)
assert len(out) == 7
model.train()
# eval_3d-style use: fresh Generator, load the checkpoint written above, eval forward
g = Generator({'input_dim': 1, 'ngf': 16}, True)
g.load_state_dict(torch.load('/tmp/hv_dropin/t/latest_net_G.pth', map_location='cpu'))
g.eval(); g.to('cuda:0')
with torch.no_grad():
    o2 = g(model.real_A[:1], model.mask[:1], 1 - model.CAM[:1], model.slice_ratio[:1])
assert o2[3].shape == (1, 1, 256, 256)
print('dropin ok')
'''


def test_models_package_is_a_drop_in():
    env = dict(os.environ)
    env['PYTHONPATH'] = os.path.join(ROOT, 'healthivert-gan_amd') + os.pathsep + env.get('PYTHONPATH', '')
    env['HV_PRECISION'] = 'fp32'
    r = subprocess.run([sys.executable, '-c', SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=600, cwd='/tmp')
    assert r.returncode == 0 and 'dropin ok' in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
