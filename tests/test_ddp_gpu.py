"""Data-parallel ranks on one MI355X (gloo transport, device tensors) against the CPU oracle's data-parallel step: every
parameter gradient equals the MEAN of the ranks' oracle gradients and the weights stay identical on all ranks -- the
semantics of SURVEY.md section 8e (per-rank batch quirks, mean of gradients) -- in the exact-fp32 mode AND in the benchmarked
fp16 mode (gradient scale -> unscale -> all-reduce -> guarded Adam), with an fp16 overflow seen by ONE rank only, and at BASELINE
config #3's per-rank shape (bs 16, full-size discriminators).  Plus the RCCL call sequence -- collectives captured inside the
step's hipGraph -- in a one-rank RCCL group."""
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
rank, port, dst, prec, world, bs, ndf, steps = int(sys.argv[1]), sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7]), int(sys.argv[8])
batch_of = rank %% int(sys.argv[9]) if len(sys.argv) > 9 else rank      # (ranks r and r + m train on the same batch)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HV_PRECISION=prec)
dist.init_process_group('gloo', rank=rank, world_size=world)
import hvgan
from hvgan import synth, ddp
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(11 + 100 * rank)      # DIFFERENT initial weights per rank: the model must broadcast rank 0's
model = Pix2PixModel(make_opt(ndf=ndf))
nets = ('G', 'D_1', 'D_2', 'D_3')
snap = lambda: {n: {k: v.detach().cpu().clone() for k, v in getattr(model, 'net' + n).state_dict().items()} for n in nets}
out = {'w0': snap()}
for step in range(steps):      # steps 3 and 4 replay the captured hipGraphs with the reductions between them (gloo cannot be captured)
    model.set_input(synth.make_batch(bs, 256, seed=100 + batch_of + 10 * step))
    model.optimize_parameters()
    if step == 0:
        torch.cuda.synchronize()
        out['g1'] = {n: {k: p.grad.detach().cpu().clone() for k, p in getattr(model, 'net' + n).named_parameters()} for n in nets}
        out['w1'] = snap()
        out['l1'] = dict(model.get_current_losses())
        out['fake_mask'] = model.fake_B_mask_raw.detach().cpu().clone()
torch.cuda.synchronize()
if steps > 2:
    assert model._graphs is not None and len(model._graphs) == 3 and model.dp_schedule == 'graphs' and not model._inline_exchange
out['wN'] = snap()
out['lN'] = dict(model.get_current_losses())
out['overflow'] = model.overflow_steps()
torch.save(out, dst + '/rank%%d.pt' %% rank)
dist.destroy_process_group()
print('ok', rank)
'''


def _rel(a, b):
    return float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-12))


def _wait_ranks(procs, timeout):
    """Wait for the rank processes; a rank that dies leaves the others waiting in a collective, so on the first failure or timeout the rest is killed
    and every rank's output is shown."""
    import time
    t0, outs = time.time(), {}
    while len(outs) < len(procs):
        for i, p in enumerate(procs):
            if i not in outs and p.poll() is not None:
                outs[i] = p.stdout.read().decode()
        bad = [i for i in outs if procs[i].returncode != 0]
        if bad or time.time() - t0 > timeout:
            for i, p in enumerate(procs):
                if i not in outs:
                    p.kill()
                    outs[i] = 'KILLED (still running after %.0f s)\n' % (time.time() - t0) + p.stdout.read().decode()
            raise AssertionError('\n'.join('--- rank %d (rc %s)\n%s' % (i, procs[i].returncode, outs[i][-2500:]) for i in sorted(outs)))
        time.sleep(0.5)
    for i in sorted(outs):
        assert b'ok' in outs[i].encode(), outs[i][-2500:]


def _run_ranks(tmp_path, world, prec, bs, ndf, steps, port_base, script=RANK, extra=()):
    port = str(port_base + os.getpid() % 1000)
    procs = [subprocess.Popen([sys.executable, '-c', script % (ROOT, ROOT), str(r), port, str(tmp_path), prec, str(world), str(bs), str(ndf), str(steps)] + list(extra),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    _wait_ranks(procs, 600)
    return [torch.load(tmp_path / ('rank%d.pt' % r)) for r in range(world)]


def _same_weights_everywhere(ranks, tags):
    a = ranks[0]
    for b in ranks[1:]:
        for tag in tags:
            for n in a[tag]:
                for k in a[tag][n]:
                    if 'running' in k or 'tracked' in k:
                        continue      # BatchNorm running statistics are per-rank data statistics (saved from rank 0)
                    assert torch.equal(a[tag][n][k], b[tag][n][k]), (tag, n, k)
        for n in a['g1']:
            for k in a['g1'][n]:
                assert torch.equal(a['g1'][n][k], b['g1'][n][k]), (n, k)


@pytest.mark.parametrize('precision', ['fp32', 'fp16'])
def test_two_ranks_equal_the_mean_of_oracle_gradients(tmp_path, precision):
    """the N-rank result == one step whose gradients are the MEAN of the N single-rank oracle gradients.
    Two ranks (gloo transport, device tensors, one MI355X), different batches per rank, four steps (the last two replay the step's three
    captured graphs with the gradient means between them), in the exact-fp32 mode (<= 2e-3 relative L2 per tensor) and in the BENCHMARKED fp16
    mode (scaled seeds -> unscale -> mean over the ranks -> guarded Adam; the fp16 gates of test_step_gpu: matrices <= 6 %, pixel-sum vectors
    <= 16 %; D_2, whose input is a thresholded mask that fp16 rounding may flip, is held to the cross-rank identity only):
      * rank 1's different seed is overridden by the broadcast of rank 0's initial weights;
      * after step 1 every parameter's .grad on both ranks equals the mean of the two oracle ranks' gradients (a missing 1/world_size would
        show as a factor 2) and the losses are each rank's own;
      * weights stay bit-identical across the ranks through all four steps and follow the oracle's data-parallel weights; no step is skipped."""
    a, b = _run_ranks(tmp_path, 2, precision, 2, 16, 4, 29600 + (1000 if precision == 'fp16' else 0))
    _same_weights_everywhere([a, b], ('w0', 'w1', 'wN'))
    assert a['overflow'] == b['overflow'] == {n: 0 for n in ('G', 'D_1', 'D_2', 'D_3')}
    # ---- the oracle's data-parallel steps from the same initial weights
    import hvgan  # noqa: F401
    from hvgan import synth
    from oracle import restate as R
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 8) // 2)))
    names = ('D_1', 'D_2', 'D_3')
    mk = lambda: R.StepState(a['w0']['G'], [a['w0'][n] for n in names], lr=2e-4, beta1=0.5, norm='batch', gan_mode='vanilla', lambda_l1=200.0)
    st = [mk(), mk()]
    fp32 = precision == 'fp32'
    gtol, vtol, ltol = (2e-3, 2e-3, 2e-3) if fp32 else (6e-2, 1.6e-1, 6e-3)
    for step in range(4):
        res = R.pix2pix_step_data_parallel(st, [synth.to_model_inputs(synth.make_batch(2, 256, seed=100 + r + 10 * step)) for r in range(2)])
        if step == 0:
            for r, got in ((0, a), (1, b)):
                for k, v in res[r][0].items():
                    # fp16 mode: the losses behind the fine_seg > 0.5 threshold (edge, D_2's) move with the pixels that fp16 rounding flips (test_step_gpu)
                    tol = ltol if (fp32 or k not in ('edge', 'D_real_2', 'D_fake_2')) else 2e-2
                    assert abs(got['l1'][k] - v) <= tol * max(1.0, abs(v)), ('loss', r, k, got['l1'][k], v)
            for k in st[0].g_params:
                e = _rel(a['g1']['G'][k], st[0].g[k].grad)
                assert e <= (gtol if a['g1']['G'][k].dim() > 1 else vtol), ('G grad', k, e)
            for d, n in enumerate(names):
                if not fp32 and n == 'D_2':
                    continue
                for k in st[0].d_params[d]:
                    e = _rel(a['g1'][n][k], st[0].d[d][k].grad)
                    assert e <= (gtol if a['g1'][n][k].dim() > 1 else vtol), (n, k, e)
            for k in st[0].g_params:      # one Adam step moves every weight by at most lr = 2e-4
                assert (a['w1']['G'][k] - st[0].g[k].detach()).abs().max() <= 4.1e-4, k
    # after four steps: losses of each rank's last batch and the parameter norms follow the oracle
    for r, got in ((0, a), (1, b)):
        for k, v in res[r][0].items():
            tol = 5e-2 if (k.startswith('D_') or k == 'G_GAN' or (k == 'edge' and not fp32)) else (1e-2 if fp32 else 2e-2)
            assert abs(got['lN'][k] - v) <= tol * max(1.0, abs(v)), ('loss4', r, k, got['lN'][k], v)
    for k in st[0].g_params:
        ref = st[0].g[k].detach()
        assert abs(float(a['wN']['G'][k].norm()) - float(ref.norm())) <= (2e-3 if fp32 else 4e-3) * max(float(ref.norm()), 1e-3), k


OVERFLOW_RANK = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
rank, port, dst = int(sys.argv[1]), sys.argv[2], sys.argv[3]
# (no schedule preflight here: it would capture the step's graphs at step 0, with rank 1's absurd gradient scale baked in as a kernel argument)
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=port, HV_PRECISION='fp16', HV_DP_PREFLIGHT='0')
dist.init_process_group('gloo', rank=rank, world_size=2)
import hvgan
from hvgan import synth
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(31)
model = Pix2PixModel(make_opt(ndf=16))
nets = ('G', 'D_1', 'D_2', 'D_3')
snap = lambda: {n: torch.cat([p.detach().flatten().cpu().clone() for p in getattr(model, 'net' + n).parameters()]) for n in nets}
out = {'w0': snap()}
assert model.grad_scale == 8192.0
for step in range(4):
    # step 0: an absurd gradient scale on RANK 1 ONLY -- its scaled activation gradients overflow its fp16 buffers, rank 0's do not
    model.grad_scale = float(2 ** 40) if (step == 0 and rank == 1) else 8192.0
    model.set_input(synth.make_batch(2, 256, seed=300 + rank + 10 * step))
    model.optimize_parameters()
    torch.cuda.synchronize()
    out['w%%d' %% (step + 1)] = snap()
    out['of%%d' %% (step + 1)] = model.overflow_steps()
    if step == 0:
        out['finite_own'] = {n: bool(torch.isfinite(getattr(model, 'net' + n).paramset().flat_grad).all().item()) for n in nets}
out['steps'] = {n: float(getattr(model, 'optimizer_' + n)._step[0].item()) for n in nets}
torch.save(out, dst + '/rank%%d.pt' %% rank)
dist.destroy_process_group()
print('ok', rank)
'''


def test_fp16_overflow_on_one_rank_is_skipped_by_every_rank(tmp_path):
    """the overflow guard reads the flat gradient AFTER the all-reduce, so an inf / nan produced on one rank reaches every
    rank's copy and 'every rank decides alike'.  Step 1 overflows on rank 1 only (gradient scale 2^40 there): both ranks must skip all four Adam
    steps (weights bit-identical to the start, counters 1 on both), steps 2-4 (the last two as graph replays) update every network on both
    ranks, and the weights stay bit-identical ACROSS the ranks throughout."""
    port = str(31600 + os.getpid() % 1000)
    procs = [subprocess.Popen([sys.executable, '-c', OVERFLOW_RANK % (ROOT, ROOT), str(r), port, str(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    _wait_ranks(procs, 300)
    a, b = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    nets = ('G', 'D_1', 'D_2', 'D_3')
    for n in nets:
        assert not a['finite_own'][n] and not b['finite_own'][n], ('the reduced gradient must carry the overflow on BOTH ranks', n, a['finite_own'], b['finite_own'])
        for tag in ('w0', 'w1', 'w2', 'w3', 'w4'):
            assert torch.equal(a[tag][n], b[tag][n]), (tag, n)
        assert torch.equal(a['w0'][n], a['w1'][n]), ('skipped step moved the weights', n)
        assert (a['w2'][n] - a['w1'][n]).abs().max().item() > 1e-5 and torch.isfinite(a['w4'][n]).all(), n
    for tag in ('of1', 'of2', 'of3', 'of4'):
        assert a[tag] == b[tag] == {n: 1 for n in nets}, (tag, a[tag], b[tag])
    assert a['steps'] == b['steps'] == {n: 3.0 for n in nets}, (a['steps'], b['steps'])


def test_config3_shape_four_ranks_bs16_equal_the_mean_of_oracle_gradients(tmp_path):
    """BASELINE config #3 at its per-rank shape ('N-rank result == average of N single-rank bs = 16 oracle gradients'): FOUR
    ranks x bs 16, 256 x 256, full-size discriminators (ndf 64), exact-fp32 mode, one step; every parameter gradient on every rank equals the mean
    of the four ranks' oracle gradients (<= 2e-3 relative L2) and each rank's losses are its own batch's.  Ranks 2 and 3 train on the batches of ranks
    0 and 1, so the mean over the four equals the mean over two oracle ranks (two bs-16 oracle steps instead of four: the CPU side of this test is
    what takes the time) while a wrong divisor, a rank left out of the collective or a pairwise-only exchange would still show.  (Eight ranks would
    exceed the GPU box's limit of six processes on the card; the arithmetic -- flat mean over world_size -- does not depend on N.)"""
    world = 4
    ranks = _run_ranks(tmp_path, world, 'fp32', 16, 64, 1, 32600, extra=('2',))
    _same_weights_everywhere(ranks, ('w0', 'w1'))
    import hvgan  # noqa: F401
    from hvgan import synth
    from oracle import restate as R
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 8) // 2)))
    a = ranks[0]
    names = ('D_1', 'D_2', 'D_3')
    st = [R.StepState(a['w0']['G'], [a['w0'][n] for n in names], lr=2e-4, beta1=0.5, norm='batch', gan_mode='vanilla', lambda_l1=200.0) for _ in range(2)]
    res = R.pix2pix_step_data_parallel(st, [synth.to_model_inputs(synth.make_batch(16, 256, seed=100 + r)) for r in range(2)])
    for r in range(world):
        for k, v in res[r % 2][0].items():
            assert abs(ranks[r]['l1'][k] - v) <= 2e-3 * max(1.0, abs(v)), ('loss', r, k, ranks[r]['l1'][k], v)
    for k in st[0].g_params:
        e = _rel(a['g1']['G'][k], st[0].g[k].grad)
        assert e <= 2e-3, ('G grad', k, e)
    for d, n in enumerate(names):
        for k in st[0].d_params[d]:
            e = _rel(a['g1'][n][k], st[0].d[d][k].grad)
            assert e <= 2e-3, (n, k, e)


RCCL_ONE = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=sys.argv[1], HV_PRECISION=sys.argv[5], HV_DDP_FORCE=sys.argv[2], HV_DP_SCHEDULE=sys.argv[4])
torch.cuda.set_device(0)
if sys.argv[2] == '1':
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))     # "nccl" IS RCCL on ROCm
import hvgan
from hvgan import synth, ddp
from hvgan.models.pix2pix_model import Pix2PixModel
from test_step_gpu import make_opt
torch.manual_seed(11)
model = Pix2PixModel(make_opt(ndf=16))
model.strict_graph = True
assert ddp.GradSync.active() == (sys.argv[2] == '1')
for step in range(5):
    model.set_input(synth.make_batch(2, 256, seed=100 + 10 * step))
    model.optimize_parameters()
torch.cuda.synchronize()
assert model._graphs is not None
if sys.argv[2] == '1':
    if sys.argv[4] == 'auto':          # the preflight ran both schedules, both correct, kept one, and put the weights back (checked by the caller: bit-identity)
        rec = model.dp_preflight_record
        assert rec and rec['chosen'] == model.dp_schedule and model.dp_schedule in ('captured', 'overlapped', 'graphs') and len(rec['schedules']) == 3, rec
        assert all(r['ok'] and r['weights_identical_across_ranks'] and r['error'] is None and r['ms_per_step'] > 0 for r in rec['schedules'].values()), rec
        assert len(model._graphs) == (3 if model.dp_schedule == 'graphs' else 1)
    elif sys.argv[4] in ('captured', 'overlapped'):      # the collectives are INSIDE the one step graph
        assert len(model._graphs) == 1 and model._inline_exchange and getattr(model, 'dp_capture_error', None) is None, (len(model._graphs), getattr(model, 'dp_capture_error', None))
    else:
        assert len(model._graphs) == 3 and not model._inline_exchange
    assert model.grad_sync.rccl is not None      # both schedules talk to RCCL through our own communicator
    model.grad_sync.close()
    t = torch.tensor([1.5], device='cuda:0', dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
    assert float(t.item()) == 1.5
    dist.destroy_process_group()
else:
    assert len(model._graphs) == 1
sd = {n: {k: v.detach().cpu() for k, v in getattr(model, 'net' + n).state_dict().items()} for n in ('G', 'D_1', 'D_3')}
torch.save(sd, sys.argv[3])
print('ok')
'''


@pytest.mark.parametrize('schedule,precision', [('captured', 'fp16'), ('captured', 'fp32'), ('overlapped', 'fp16'), ('graphs', 'fp32'), ('auto', 'fp16')])
def test_rccl_exchange_path_single_rank(tmp_path, schedule, precision):
    """The gradient exchange exactly as a multi-GPU job issues it -- ncclAllReduce (ncclAvg) of the flat gradient buffers through a communicator of our own
    (ddp.RcclComm): between the step's three graphs (HV_DP_SCHEDULE=graphs), CAPTURED INSIDE the step's one hipGraph on its main branch (captured: the three
    discriminators' arena where their streams join, G's before its Adam step), or captured per discriminator and chained D_1 -> D_2 -> D_3 -> G by events
    (overlapped: D_k's mean beside the other discriminators' passes) -- plus broadcast, barrier and
    the MAX-reduce of the bench clock, in a one-rank RCCL group: averaging over one rank is the identity, so the weights after five steps must equal those of a
    run without a process group, bit for bit.  'auto' (the default of a multi-GPU job) first runs the PREFLIGHT -- both schedules, 2 x 10 steps, cross-rank
    checks -- which must leave the weights, running statistics and Adam state exactly as it found them: the same bit-identity after the five steps."""
    port = str(29700 + os.getpid() % 1000 + {'graphs': 1000, 'auto': 2000, 'overlapped': 3000}.get(schedule, 0) + (500 if precision == 'fp16' else 0))
    outs = []
    for force in ('1', '0'):
        dst = str(tmp_path / ('w%s.pt' % force))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
        p = subprocess.run([sys.executable, '-c', RCCL_ONE % (ROOT, ROOT), port, force, dst, schedule, precision], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                           timeout=600, env=env)
        assert p.returncode == 0 and b'ok' in p.stdout, p.stdout.decode()[-3000:]
        outs.append(torch.load(dst))
    for n in outs[0]:
        for k in outs[0][n]:
            assert torch.equal(outs[0][n][k], outs[1][n][k]), (n, k)


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


def test_bench_two_ranks_rehearsal_over_gloo():
    """`python bench.py --gpus 2` from a plain shell, the whole flow the driver's scaling run takes -- launcher, process group from the
    environment, broadcast, the data-parallel step, survey / timed / roofline legs, MAX-over-ranks clock, one JSON line from rank 0 -- with the
    gloo transport so that both ranks can share this box's one GPU (RCCL refuses two ranks per device; gloo cannot be captured, so the step is cut
    into its three graphs with the means between them)."""
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
    assert rec['comm']['backend'] == 'gloo' and rec['comm']['world_size'] == 2 and rec['scaling'] == 'weak' and rec['value'] > 0
    pf = rec['comm']['preflight']      # (gloo: one schedule to try; the preflight still proves cross-rank weight identity and puts the weights back)
    assert pf and pf['chosen'] == 'graphs' and pf['schedules']['graphs']['ok'] and pf['schedules']['graphs']['weights_identical_across_ranks'], rec['comm']
    assert rec['roofline'] and rec['roofline']['launches'] > 0 and 'fine_generator_forward' in rec


def test_bench_data_parallel_dress_rehearsal_on_one_device():
    """The driver's multi-GPU bench without a node: the real bench with the data-parallel step in a one-rank RCCL group -- the default cut schedule (three
    graphs, the means between them through torch.distributed) and the captured one (direct ncclAllReduce calls inside the one step graph) --, and the
    same started as TWO ranks by torch.distributed.run on this one device (gloo transport): the JSON line must report
    what the collective layer saw (n_gpus, global batch, backend, world size, schedule, graphs per step).  The step-time ratio against the
    single-process step is printed, and asserted only under HV_PERF_ASSERT=1 (wall-clock ratios do not belong in a correctness suite: tools/dp_ratio.sh)."""
    plain = _bench({}, ['--steps', '10', '--warmup', '3'])
    assert plain['n_gpus'] == 1 and plain['comm']['world_size'] == 1 and '1 graphs' in plain['config']['launch'], plain['config']      # (single process: the whole step is one graph)
    assert len(plain['regions_ms_per_step']) == 3 and plain['ms_per_step'] == sorted(plain['regions_ms_per_step'])[1]
    one = _bench({'HV_DDP_FORCE': '1', 'HV_DP_SCHEDULE': 'graphs'}, ['--steps', '10', '--warmup', '3'])
    assert one['comm']['backend'] == 'nccl' and one['comm']['world_size'] == 1 and one['comm']['dp_schedule'] == 'graphs', one['comm']
    assert '3 graphs' in one['config']['launch'], (one['comm'], one['config'])
    # what the driver's `bench.py --gpus N` takes (no schedule given): the preflight picks, and the line says what it saw
    auto = _bench({'HV_DDP_FORCE': '1'}, ['--steps', '10', '--warmup', '3'])
    pf = auto['comm']['preflight']
    assert pf and pf['chosen'] == auto['comm']['dp_schedule'] and all(r['ok'] for r in pf['schedules'].values()), auto['comm']
    print('preflight in a one-rank RCCL group: %s' % {k: v['ms_per_step'] for k, v in pf['schedules'].items()}, '->', pf['chosen'])
    cap = _bench({'HV_DDP_FORCE': '1', 'HV_DP_SCHEDULE': 'captured'}, ['--steps', '10', '--warmup', '3'])
    assert '1 graphs' in cap['config']['launch'] and cap['comm']['dp_schedule'] == 'captured' and cap['comm']['capture_error'] is None, (cap['comm'], cap['config'])
    print('data-parallel schedule in a one-rank RCCL group: cut (default) %.3f ms, captured %.3f ms, single process %.3f ms' % (one['ms_per_step'], cap['ms_per_step'], plain['ms_per_step']))
    if os.environ.get('HV_PERF_ASSERT') == '1':
        assert cap['ms_per_step'] <= 1.03 * plain['ms_per_step'], ('captured data-parallel step vs single-process step', cap['ms_per_step'], plain['ms_per_step'])
    two = _bench({'HV_DDP_BACKEND': 'gloo'}, ['--gpus', '2', '--steps', '4', '--warmup', '3'],
                 launcher=[sys.executable, '-m', 'torch.distributed.run', '--standalone', '--nnodes=1', '--nproc-per-node', '2', '--local-addr', '127.0.0.1'])
    assert two['n_gpus'] == 2 and two['config']['global_batch'] == 32 and two['config']['parallelism'] == 'dp2', two['config']
    assert two['comm']['backend'] == 'gloo' and two['comm']['world_size'] == 2, two['comm']
    assert two['value'] > 0 and two['scaling'] == 'weak' and '3 graphs' in two['config']['launch']
