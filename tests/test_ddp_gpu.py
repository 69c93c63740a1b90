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


RCCL_ONE = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], HV_PRECISION='fp32', HV_DDP_FORCE=sys.argv[2])
torch.cuda.set_device(0)
if sys.argv[2] == '1':
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))     # "nccl" IS RCCL on ROCm
import hvgan
from hvgan import synth, ddp
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(11)
model = Pix2PixModel(make_opt(ndf=16))
ddp.broadcast_parameters([model.netG, model.netD_1, model.netD_2, model.netD_3])
assert ddp.GradSync.active() == (sys.argv[2] == '1')
for step in range(4):
    model.set_input(synth.make_batch(2, 256, seed=100 + 10 * step))
    model.optimize_parameters()
torch.cuda.synchronize()
assert model._graphs is not None
if sys.argv[2] == '1':
    t = torch.tensor([1.5], device='cuda:0', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
    assert float(t.item()) == 1.5
    dist.destroy_process_group()
sd = {n: {k: v.detach().cpu() for k, v in getattr(model, 'net' + n).state_dict().items()} for n in ('G', 'D_1')}
torch.save(sd, sys.argv[3])
print('ok')
'''


def test_rccl_exchange_path_single_rank(tmp_path):
    """The gradient exchange exactly as a multi-GPU job issues it (RCCL all-reduce of the flat gradient buffers on the side stream between the
    captured step graphs, broadcast, barrier, MAX-reduce of the bench clock) in a one-rank RCCL group: averaging over one rank is the
    identity, so the weights after four steps must equal those of a run without a process group, bit for bit."""
    port = str(29700 + os.getpid() % 1000)
    outs = []
    for force in ('1', '0'):
        dst = str(tmp_path / ('w%s.pt' % force))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        p = subprocess.run([sys.executable, '-c', RCCL_ONE % (ROOT, ROOT), port, force, dst], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=600, env=env)
        assert p.returncode == 0 and b'ok' in p.stdout, p.stdout.decode()[-3000:]
        outs.append(torch.load(dst))
    for n in outs[0]:
        for k in outs[0][n]:
            assert torch.equal(outs[0][n][k], outs[1][n][k]), (n, k)
