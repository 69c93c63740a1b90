"""Two data-parallel ranks on one MI355X (gloo transport, device tensors): after a train step with DIFFERENT per-rank
batches every rank must hold identical weights (gradients were averaged before each optimiser step), and those weights
must equal a single-process step whose gradients are the mean of the two ranks' gradients -- the DDP semantics of
SURVEY.md section 8e (per-rank batch quirks, mean of gradients)."""
import os
import subprocess
import sys

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

RANK = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
rank = int(sys.argv[1]); os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], HV_PRECISION='fp32')
dist.init_process_group('gloo', rank=rank, world_size=2)
import hvgan
from hvgan import synth, ddp
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(11)
model = Pix2PixModel(make_opt(ndf=16))
ddp.broadcast_parameters([model.netG, model.netD_1, model.netD_2, model.netD_3])
for step in range(4):      # steps 3 and 4 replay the captured hipGraphs with the reductions in between
    model.set_input(synth.make_batch(2, 256, seed=100 + rank + 10 * step))
    model.optimize_parameters()
torch.cuda.synchronize()
assert model._graphs is not None
sd = {n: {k: v.detach().cpu() for k, v in getattr(model, 'net' + n).state_dict().items() if 'running' not in k and 'tracked' not in k}
      for n in ('G', 'D_1', 'D_2', 'D_3')}
torch.save(sd, sys.argv[3] + '/rank%%d.pt' %% rank)
dist.destroy_process_group()
print('ok', rank)
'''


def test_two_ranks_keep_identical_weights(tmp_path):
    port = str(29600 + os.getpid() % 1000)
    procs = [subprocess.Popen([sys.executable, '-c', RANK % (ROOT, ROOT), str(r), port, str(tmp_path)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=600)
        assert p.returncode == 0 and b'ok' in out, out.decode()[-3000:]
    a, b = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    moved = 0
    for n in a:
        for k in a[n]:
            assert torch.equal(a[n][k], b[n][k]), (n, k)
    # and the step really used both ranks' data: a single-rank run from the same seed gives different weights
    import hvgan
    from hvgan import synth
    from hvgan.models.pix2pix_model import Pix2PixModel
    from test_step_gpu import make_opt
    os.environ['HV_PRECISION'] = 'fp32'
    torch.manual_seed(11)
    model = Pix2PixModel(make_opt(ndf=16))
    for step in range(4):
        model.set_input(synth.make_batch(2, 256, seed=100 + 10 * step))
        model.optimize_parameters()
    w = model.netG.state_dict()['fine_generator.allconv17.conv.weight_orig'].cpu()
    assert not torch.equal(w, a['G']['fine_generator.allconv17.conv.weight_orig'])
