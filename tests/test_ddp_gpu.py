"""Two data-parallel ranks on one MI355X (gloo transport, device tensors) against the CPU oracle's data-parallel step: every
parameter gradient equals the MEAN of the two ranks' oracle gradients and the weights stay identical on both ranks -- the
semantics of SURVEY.md section 8e (per-rank batch quirks, mean of gradients).  Plus the RCCL call sequence in a one-rank group."""
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
rank = int(sys.argv[1]); os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[2], HV_PRECISION='fp32', HV_DP_SCHEDULE=sys.argv[4])
dist.init_process_group('gloo', rank=rank, world_size=2)
import hvgan
from hvgan import synth, ddp
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(11 + 100 * rank)      # DIFFERENT initial weights per rank: the model must broadcast rank 0's
model = Pix2PixModel(make_opt(ndf=16))
nets = ('G', 'D_1', 'D_2', 'D_3')
snap = lambda: {n: {k: v.detach().cpu().clone() for k, v in getattr(model, 'net' + n).state_dict().items()} for n in nets}
out = {'w0': snap()}
for step in range(4):      # steps 3 and 4 replay the captured hipGraphs with the reductions in between
    model.set_input(synth.make_batch(2, 256, seed=100 + rank + 10 * step))
    model.optimize_parameters()
    if step == 0:
        torch.cuda.synchronize()
        out['g1'] = {n: {k: p.grad.detach().cpu().clone() for k, p in getattr(model, 'net' + n).named_parameters()} for n in nets}
        out['w1'] = snap()
        out['l1'] = dict(model.get_current_losses())
torch.cuda.synchronize()
if sys.argv[4] == 'phases':
    assert model._dp_graphs is not None and len(model._dp_graphs) == 12
else:
    assert model._graphs is not None and len(model._graphs) == 3 and not model._dp_graphs
out['w4'] = snap()
out['l4'] = dict(model.get_current_losses())
torch.save(out, sys.argv[3] + '/rank%%d.pt' %% rank)
dist.destroy_process_group()
print('ok', rank)
'''


def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-12))


@pytest.mark.parametrize('schedule', ['graphs', 'phases'])
def test_two_ranks_equal_the_mean_of_oracle_gradients(tmp_path, schedule):
    """the N-rank result == one step whose gradients are the MEAN of the N single-rank oracle gradients.
    Two ranks (gloo transport, device tensors, one MI355X), fp32 parity mode, different batches per rank, four steps (the last two
    replay the captured graphs -- the single-process step's three with the gradient means between them (default), or the twelve phase
    graphs of HV_DP_SCHEDULE=phases -- with the exchanges between them):
      * rank 1's different seed is overridden by the broadcast of rank 0's initial weights;
      * after step 1 every parameter's .grad on both ranks equals the mean of the two oracle ranks' gradients (<= 2e-3 relative L2,
        a missing 1/world_size would show as a factor 2) and the losses are each rank's own;
      * weights stay bit-identical across the ranks through all four steps and follow the oracle's data-parallel weights."""
    port = str(29600 + os.getpid() % 1000 + (1000 if schedule == 'phases' else 0))
    procs = [subprocess.Popen([sys.executable, '-c', RANK % (ROOT, ROOT), str(r), port, str(tmp_path), schedule], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    for p in procs:
        out, _ = p.communicate(timeout=900)
        assert p.returncode == 0 and b'ok' in out, out.decode()[-3000:]
    a, b = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    for tag in ('w0', 'w1', 'w4'):
        for n in a[tag]:
            for k in a[tag][n]:
                if 'running' in k or 'tracked' in k:
                    continue      # BatchNorm running statistics are per-rank data statistics (saved from rank 0)
                assert torch.equal(a[tag][n][k], b[tag][n][k]), (tag, n, k)
    for n in a['g1']:
        for k in a['g1'][n]:
            assert torch.equal(a['g1'][n][k], b['g1'][n][k]), (n, k)
    # ---- the oracle's data-parallel steps from the same initial weights
    import hvgan  # noqa: F401
    from hvgan import synth
    from oracle import restate as R
    torch.set_num_threads(max(1, (os.cpu_count() or 8) // 2))
    names = ('D_1', 'D_2', 'D_3')
    mk = lambda: R.StepState(a['w0']['G'], [a['w0'][n] for n in names], lr=2e-4, beta1=0.5, norm='batch', gan_mode='vanilla', lambda_l1=200.0)
    st = [mk(), mk()]
    for step in range(4):
        res = R.pix2pix_step_data_parallel(st, [synth.to_model_inputs(synth.make_batch(2, 256, seed=100 + r + 10 * step)) for r in range(2)])
        if step == 0:
            for r, got in ((0, a), (1, b)):
                for k, v in res[r][0].items():
                    assert abs(got['l1'][k] - v) <= 2e-3 * max(1.0, abs(v)), ('loss', r, k, got['l1'][k], v)
            worst = 0.0
            for k in st[0].g_params:
                e = _rel(a['g1']['G'][k], st[0].g[k].grad)
                worst = max(worst, e)
                assert e <= 2e-3, ('G grad', k, e)
            for d, n in enumerate(names):
                for k in st[0].d_params[d]:
                    e = _rel(a['g1'][n][k], st[0].d[d][k].grad)
                    assert e <= 2e-3, (n, k, e)
            for k in st[0].g_params:      # one Adam step moves every weight by at most lr = 2e-4
                assert (a['w1']['G'][k] - st[0].g[k].detach()).abs().max() <= 4.1e-4, k
    # after four steps: losses of each rank's last batch and the parameter norms follow the oracle
    for r, got in ((0, a), (1, b)):
        for k, v in res[r][0].items():
            tol = 5e-2 if k.startswith('D_') or k == 'G_GAN' else 1e-2
            assert abs(got['l4'][k] - v) <= tol * max(1.0, abs(v)), ('loss4', r, k, got['l4'][k], v)
    for k in st[0].g_params:
        ref = st[0].g[k].detach()
        assert abs(float(a['w4']['G'][k].norm()) - float(ref.norm())) <= 2e-3 * max(float(ref.norm()), 1e-3), k


RCCL_ONE = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], HV_PRECISION='fp32', HV_DDP_FORCE=sys.argv[2], HV_DP_SCHEDULE=sys.argv[4])
if sys.argv[4] == 'phases':
    os.environ['HV_BATCH_D'] = '0'      # the twelve-phase schedule always takes the split real-first discriminator passes: same summation order for the bit-for-bit comparison
torch.cuda.set_device(0)
if sys.argv[2] == '1':
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))     # "nccl" IS RCCL on ROCm
import hvgan
from hvgan import synth, ddp
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(11)
model = Pix2PixModel(make_opt(ndf=16))
assert ddp.GradSync.active() == (sys.argv[2] == '1')
for step in range(4):
    model.set_input(synth.make_batch(2, 256, seed=100 + 10 * step))
    model.optimize_parameters()
torch.cuda.synchronize()
assert (model._dp_graphs if (sys.argv[2] == '1' and sys.argv[4] == 'phases') else model._graphs) is not None
if sys.argv[2] == '1':
    t = torch.tensor([1.5], device='cuda:0', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
    assert float(t.item()) == 1.5
    dist.destroy_process_group()
sd = {n: {k: v.detach().cpu() for k, v in getattr(model, 'net' + n).state_dict().items()} for n in ('G', 'D_1')}
torch.save(sd, sys.argv[3])
print('ok')
'''


@pytest.mark.parametrize('schedule', ['graphs', 'phases'])
def test_rccl_exchange_path_single_rank(tmp_path, schedule):
    """The gradient exchange exactly as a multi-GPU job issues it (RCCL all-reduce of the flat gradient buffers on the side stream between the
    captured step graphs, broadcast, barrier, MAX-reduce of the bench clock) in a one-rank RCCL group: averaging over one rank is the
    identity, so the weights after four steps must equal those of a run without a process group, bit for bit."""
    port = str(29700 + os.getpid() % 1000 + (1000 if schedule == 'phases' else 0))
    outs = []
    for force in ('1', '0'):
        dst = str(tmp_path / ('w%s.pt' % force))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        p = subprocess.run([sys.executable, '-c', RCCL_ONE % (ROOT, ROOT), port, force, dst, schedule], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=600, env=env)
        assert p.returncode == 0 and b'ok' in p.stdout, p.stdout.decode()[-3000:]
        outs.append(torch.load(dst))
    for n in outs[0]:
        for k in outs[0][n]:
            assert torch.equal(outs[0][n][k], outs[1][n][k]), (n, k)


def test_bench_two_ranks_rehearsal_over_gloo():
    """`python bench.py --gpus 2` from a plain shell, the whole flow the driver's scaling run takes -- launcher, process group from the
    environment, broadcast, the data-parallel step with its twelve phase graphs, survey / timed / roofline legs, MAX-over-ranks clock, one
    JSON line from rank 0 -- with the gloo transport so that both ranks can share this box's one GPU (RCCL refuses two ranks per device)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    env.update(HV_DDP_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '3', '--no-cpu-baseline'],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['global_batch'] == 32 and rec['config']['parallelism'] == 'dp2'
    assert rec['config']['launch'].startswith('hipGraph replay (3 graphs/step)')
    assert rec['roofline'] and rec['roofline']['launches'] > 0 and 'fine_generator_forward' in rec


def _bench(extra_env, args, launcher=None):
    import json
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'HV_DDP_FORCE', 'HV_DDP_BACKEND')}
    env.update(extra_env)
    cmd = (launcher or [sys.executable]) + [os.path.join(ROOT, 'bench.py')] + args + ['--no-cpu-baseline', '--no-inference', '--no-extra']
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, env=env)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.startswith('{')]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


def test_bench_data_parallel_dress_rehearsal_on_one_device():
    """The driver's multi-GPU bench without a node: (a) the real bench with the data-parallel SCHEDULE (the step's three graphs with RCCL all-reduce
    calls on the exchange stream between them) in a one-rank RCCL group against the plain three-graph step on the same device -- the schedule itself
    must cost < 8 % over the single-process one-graph step (the twelve-phase schedule of HV_DP_SCHEDULE=phases: < 15 %, measured 7-10 % over the three-graph step);
    (b) the real bench started as TWO ranks by torch.distributed.run on this one device (gloo transport -- RCCL refuses two ranks on one device):
    the JSON line must report what the collective layer saw (n_gpus, global batch, backend, world size)."""
    plain = _bench({}, ['--steps', '10', '--warmup', '3'])
    assert plain['n_gpus'] == 1 and plain['comm']['world_size'] == 1 and '1 graphs' in plain['config']['launch'], plain['config']      # (single process: the whole step is one graph)
    assert len(plain['regions_ms_per_step']) == 3 and plain['ms_per_step'] == sorted(plain['regions_ms_per_step'])[1]
    one = _bench({'HV_DDP_FORCE': '1'}, ['--steps', '10', '--warmup', '3'])
    assert one['comm']['backend'] == 'nccl' and one['comm']['world_size'] == 1 and '3 graphs' in one['config']['launch'], (one['comm'], one['config'])
    # (measured: +0.14-0.20 ms for the two stream hops and one-rank collectives, +0.11 ms for three graphs instead of the single-process one: ~4 % of 7.7 ms)
    assert one['ms_per_step'] <= 1.08 * plain['ms_per_step'], ('data-parallel schedule vs single-rank step', one['ms_per_step'], plain['ms_per_step'])
    ph = _bench({'HV_DDP_FORCE': '1', 'HV_DP_SCHEDULE': 'phases'}, ['--steps', '10', '--warmup', '3'])
    assert '12 graphs' in ph['config']['launch'] and ph['ms_per_step'] <= 1.15 * plain['ms_per_step'], (ph['config'], ph['ms_per_step'], plain['ms_per_step'])
    two = _bench({'HV_DDP_BACKEND': 'gloo'}, ['--gpus', '2', '--steps', '4', '--warmup', '3'],
                 launcher=[sys.executable, '-m', 'torch.distributed.run', '--standalone', '--nnodes=1', '--nproc-per-node', '2', '--local-addr', '127.0.0.1'])
    assert two['n_gpus'] == 2 and two['config']['global_batch'] == 32 and two['config']['parallelism'] == 'dp2', two['config']
    assert two['comm']['backend'] == 'gloo' and two['comm']['world_size'] == 2, two['comm']
    assert two['value'] > 0 and two['scaling'] == 'weak' and '3 graphs' in two['config']['launch']
