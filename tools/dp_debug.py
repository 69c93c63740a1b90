"""One-rank RCCL rehearsal of a data-parallel schedule under faulthandler (where does a crash come from?).  python -X faulthandler tools/dp_debug.py captured"""
import os, sys, faulthandler
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29977', HV_PRECISION='fp16', HV_DDP_FORCE='1', HV_DP_SCHEDULE=sys.argv[1])
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(11)
model = Pix2PixModel(make_opt(ndf=16))
model.strict_graph = True
for step in range(5):
    print('step', step, flush=True)
    model.set_input(synth.make_batch(2, 256, seed=100 + 10 * step))
    model.optimize_parameters()
    torch.cuda.synchronize()
print('schedule', model.dp_schedule, 'graphs', len(model._graphs or ()), 'preflight', model.dp_preflight_record, flush=True)
model.grad_sync.close()
dist.destroy_process_group()
print('ok')
