#!/usr/bin/env python
"""bench.py -- sagittal slices/sec of one full HealthiVert-GAN train step (G + 3xD) at 256x256, per-GPU bs=16.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Launched by `python -m torch.distributed.run` (the driver's form) the ranks are
already there; from a plain shell this script starts them itself -- as CHILD processes, before anything in the parent
has touched the GPU -- and exits with their status.  A run whose rank count differs from --gpus exits non-zero.

One "step" = Pix2PixModel.optimize_parameters on one synthetic batch already resident in HBM: generator forward,
three discriminator updates (fake + real pass each, Adam), generator backward through D_1/D_3 + losses, Adam.
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, measured live with HIP
events on the launch stream), `fine_generator_forward` (the north-star's secondary gate), `inference` (BASELINE
config #4) and `cpu_baseline` (the CPU oracle timed on this box's host cores, rank 0, N=1 only).

    python bench.py --gpus 2 --dry-run      # CPU-only rehearsal of the multi-rank plumbing (gloo): launcher, rendezvous,
                                            # flat-gradient exchange of the real networks' sizes, MAX-over-ranks clock
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time
from argparse import Namespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_SLICE = 195.9          # SURVEY.md section 8d / BASELINE.md section 2: whole train step, per slice
GFLOP_FINE_FWD = 10.23           # FineGenerator forward per slice, the north-star's MFMA gate
GFLOP_G_FWD = 17.56              # whole Generator forward per slice
MB_FINE_FWD = 48.9               # FineGenerator forward per slice: compulsory bytes of its convolutions at fp16 storage, unfused
MFMA_PEAK_TFLOPS = {'fp16': 2500.0, 'fp32': 157.3}   # MI355X dense peaks, /opt/skills/guides/MI355X_MICROARCH.md


def make_opt(precision):
    return Namespace(gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/hv_bench', name='bench', preprocess='none', input_nc=1,
                     output_nc=1, ngf=64, ndf=64, netD='basic', netG='unet_256', n_layers_D=3, norm='batch', init_type='normal',
                     init_gain=0.02, no_dropout=True, gan_mode='vanilla', lr=2e-4, beta1=0.5, lambda_L1=200.0, direction='BtoA',
                     lr_policy='linear', epoch_count=1, n_epochs=100, n_epochs_decay=100, continue_train=False, load_iter=0,
                     epoch='latest', verbose=False, hv_precision=precision)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch (slices)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--precision', default=os.environ.get('HV_PRECISION', 'fp16'), choices=['fp16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-inference', action='store_true', help='skip the config-#4 inference sub-record')
    ap.add_argument('--no-extra', action='store_true', help='skip the device-loader and fp32-mode sub-records')
    ap.add_argument('--serial', action='store_true', help='one HIP stream for the whole run (profiling: per-kernel durations without stream-level overlap)')
    ap.add_argument('--no-graph', action='store_true', help='launch kernels eagerly instead of replaying captured hipGraphs')
    ap.add_argument('--dry-run', action='store_true', help='CPU-only rehearsal of the multi-rank path over gloo (no GPU, no kernels)')
    return ap.parse_args()


def launch_ranks(args):
    """Plain `python bench.py --gpus N`: start the N ranks as children through torch.distributed.run and hand back their
    exit status.  Nothing in this (parent) process has initialised the GPU, and it never does."""
    # --standalone: torchrun's own c10d rendezvous picks AND HOLDS a free port on this host (a port found by bind(0) / close() here could be taken
    # by another job before the ranks meet)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--local-addr', '127.0.0.1',
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.run(cmd, env=env).returncode


def dry_run(args, rank, world):
    """The multi-rank path without a GPU: gloo process group, the four networks built on the host (so the flat gradient buffers
    have the real sizes: G 3.95 MB, each D 11.05 MB), K exchange rounds through ddp.GradSync with a check of the mean, and the
    same barrier / MAX-over-ranks clock as the real run.  Prints a JSON line marked dry_run (not a performance number)."""
    import torch
    import torch.distributed as dist
    import hvgan  # noqa: F401
    from hvgan import ddp
    from hvgan.models import networks
    from hvgan.models.inpaint_networks import Generator
    if world > 1:
        os.environ['HV_DDP_BACKEND'] = 'gloo'
        ddp.init_from_env()
    torch.manual_seed(1234)
    nets = [Generator({'input_dim': 1, 'ngf': 16}, True)] + [networks.define_D(1, 64, 'basic', 3, 'batch', 'normal', 0.02, []) for _ in range(3)]
    flats = [torch.zeros(sum(p.numel() for p in n.parameters() if p.requires_grad)) for n in nets]
    gs = ddp.GradSync()

    def barrier():
        if world > 1:
            dist.barrier()
    for it in range(args.warmup + args.steps):
        if it == args.warmup:
            barrier()
            t0 = time.perf_counter()
        for i, f in enumerate(flats):
            f.fill_(float(rank + 1) * (i + 1))
            gs.reduce(f)
            want = (i + 1) * (world + 1) / 2.0          # mean over ranks of (rank+1)*(i+1)
            assert abs(float(f[0]) - want) < 1e-6 and abs(float(f[-1]) - want) < 1e-6, (float(f[0]), want)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({'metric': 'dry-run gradient exchange rounds/sec (gloo, CPU; plumbing rehearsal, not a benchmark)',
                          'value': round(args.steps / dt, 2), 'unit': 'rounds/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': 'f32', 'data': 'synthetic', 'dry_run': True,
                          'config': {'workload': 'flat-gradient mean of G + 3xD over %d gloo ranks' % world,
                                     'bytes_per_round': int(sum(f.numel() for f in flats) * 4)}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def physical_cores():
    """Physical cores of the host (distinct (physical id, core id) pairs of /proc/cpuinfo); None when unknown."""
    try:
        seen, phys = set(), None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                phys = line.split(':')[1].strip()
            elif line.startswith('core id'):
                seen.add((phys, line.split(':')[1].strip()))
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(model, size, seed, batch_size=16, warm=1, timed=3):
    """The CPU oracle (oracle/restate.py, kind 'port') on the same weights: `warm` untimed + `timed` timed full train steps at the benchmarked
    batch size (bs = 16, 1 warm-up + >= 3 timed; about two minutes on a 128-core host)."""
    import torch
    from hvgan import synth
    from oracle import restate as R
    sd_g = {k: v.detach().cpu() for k, v in model.netG.state_dict().items()}
    sd_d = [{k: v.detach().cpu() for k, v in getattr(model, 'netD_%d' % k).state_dict().items()} for k in (1, 2, 3)]
    st = R.StepState(sd_g, sd_d, lr=2e-4, beta1=0.5, norm='batch', gan_mode='vanilla', lambda_l1=200.0)
    times = []
    for i in range(warm + timed):
        batch = synth.to_model_inputs(synth.make_batch(batch_size, size, seed=seed + i))
        t0 = time.time()
        R.pix2pix_step(st, batch)
        if i >= warm:
            times.append(time.time() - t0)
    dt = sum(times) / len(times)
    return dict(value=round(batch_size / dt, 3), unit='slices/s', cores=torch.get_num_threads(), physical_cores=physical_cores(),
                logical_cpus=os.cpu_count(), kind='port',
                sample='%d warm-up + %d timed full train steps (G + 3xD) at bs=%d, %dx%d, fp32, CPU oracle oracle/restate.py, %d torch threads'
                       % (warm, timed, batch_size, size, size, torch.get_num_threads()),
                seconds_per_step=[round(t, 2) for t in times], gflops=round(GFLOP_PER_SLICE * batch_size / dt, 1))


def fp32_record(args, dev, local_rank):
    """The same step in the exact-fp32 parity mode (the reference's own arithmetic: fp32 MFMA, fp32 storage): 3 warm-up + 10 timed steps."""
    import torch
    from hvgan import synth
    from hvgan.models.pix2pix_model import Pix2PixModel
    prev = os.environ.get('HV_PRECISION')
    os.environ['HV_PRECISION'] = 'fp32'
    try:
        torch.manual_seed(1234)
        opt = make_opt('fp32')
        opt.gpu_ids = [local_rank]
        m = Pix2PixModel(opt)
        m.setup(opt)
        m.strict_graph = True
        m.set_input(synth.make_batch(args.batch, args.size, seed=1234))
        for _ in range(max(3, m.GRAPH_WARMUP + 1)):
            m.optimize_parameters()
        torch.cuda.synchronize()
        n, t0 = 10, time.perf_counter()
        for _ in range(n):
            m.optimize_parameters()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        return {'ms_per_step': round(dt * 1e3, 3), 'slices_s': round(args.batch / dt, 1), 'steps': n,
                'precision': 'fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 storage: the |d| <= 1e-3 parity mode'}
    finally:
        if prev is None:
            os.environ.pop('HV_PRECISION', None)
        else:
            os.environ['HV_PRECISION'] = prev


def config5_record(args, dev, local_rank):
    """BASELINE config #5's single-GPU workload: the same train step at 512x512, bs 16, fp16 mode (attention over 64x64 patches, L = 4096): 3 warm-up
    (incl. the capture) + 5 timed graph replays."""
    import torch
    from hvgan import synth
    from hvgan.models.pix2pix_model import Pix2PixModel
    torch.manual_seed(1234)
    opt = make_opt(args.precision)
    opt.gpu_ids = [local_rank]
    m = Pix2PixModel(opt)
    m.setup(opt)
    m.strict_graph = True
    m.set_input(synth.make_batch(16, 512, seed=1234))
    for _ in range(max(3, m.GRAPH_WARMUP + 1)):
        m.optimize_parameters()
    torch.cuda.synchronize()
    n, t0 = 5, time.perf_counter()
    for _ in range(n):
        m.optimize_parameters()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    rec = {'workload': 'the same two-stage train step at 512x512, bs 16, fp16 mode (BASELINE config #5, one GPU)', 'ms_per_step': round(dt * 1e3, 3),
           'slices_s': round(16 / dt, 1), 'steps': n, 'pixels_per_s': round(16 * 512 * 512 / dt / 1e6, 1)}
    del m
    torch.cuda.empty_cache()
    return rec


def device_loader_record(model, args, dev):
    """The real train loop: every step assembles 16 FRESH slices on the device from resident synthetic volumes (batch_assembly.DeviceBatchAssembler,
    SURVEY.md section 8f row f1) and hands them to set_input before optimize_parameters -- the bench's timed region replays one resident batch."""
    import numpy as np
    import torch
    from hvgan import synth
    from hvgan.batch_assembly import VertebraVolume, DeviceBatchAssembler
    nvol = 32
    raw = [synth.make_spine_volume(s, H=args.size, W=args.size, Z=48, pitch=48) for s in range(4)]
    vols = [VertebraVolume(*raw[i % 4], 11 + i % 3, ['10', '14'], path='v%d' % i) for i in range(nvol)]
    asm = DeviceBatchAssembler(vols, str(dev))
    rng = np.random.RandomState(0)
    order = rng.permutation(nvol)
    state = np.random.get_state()
    np.random.seed(0)

    def step(i):
        model.set_input(asm.batch([int(order[(args.batch * i + j) % nvol]) for j in range(args.batch)]))
        model.optimize_parameters()
    try:
        for i in range(4):
            step(i)
        torch.cuda.synchronize()
        n, t0 = 10, time.perf_counter()
        for i in range(n):
            step(4 + i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    finally:
        np.random.set_state(state)
    return {'ms_per_step': round(dt * 1e3, 3), 'slices_s': round(args.batch / dt, 1), 'steps': n,
            'what': 'set_input(DeviceBatchAssembler.batch(16 fresh slice draws)) + optimize_parameters per step, %d resident volumes' % nvol}


def inference_record(dev, precision):
    """BASELINE config #4: the three chained synthesis stages of one straightened 256x256x64 volume (infer.process_volume, host
    pre/post-processing and PCIe copies included), eval-mode generator with random-init weights, synthetic volume."""
    import torch
    from hvgan import synth, infer
    from hvgan.models.inpaint_networks import Generator
    torch.manual_seed(0)
    net = Generator({'input_dim': 1, 'ngf': 16}, True)
    net.fine_generator.fc_height.bias.data.fill_(0.4)        # plausible heights from untrained weights
    net.fine_generator.fc_height.weight.data.mul_(1e-2)
    net.to(dev).eval()
    net.precision = precision
    ct, label, cam = synth.make_volume(nz=64, size=256, seed=2)
    cam255 = cam * 255          # the reference scales the attention map where it loads the file (eval_3d_sagittal_twostage.py:181): input, not path
    for _ in range(2):
        infer.process_volume(net, ct, label, cam255, 20, dev)
    torch.cuda.synchronize()
    n, t0 = 5, time.perf_counter()
    for _ in range(n):
        out_ct, out_seg = infer.process_volume(net, ct, label, cam255, 20, dev)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    nz = int((out_seg.reshape(-1, out_seg.shape[2]) != 0).any(axis=0).sum())
    # steady state over a stream of volumes (infer.process_volumes: volume n + 1's pinned copy / upload / scan and volume n - 1's download overlap
    # volume n's stages): 10 volumes, the first two not counted
    def stream_rate(copy):
        vols = [(ct, label, cam255, 20)] * 10
        t, k = None, 0
        for _ in infer.process_volumes(net, vols, dev, copy=copy):
            k += 1
            if k == 2:
                t = time.perf_counter()
        return (time.perf_counter() - t) / (len(vols) - 2)
    dt_pipe, dt_pipe_views = stream_rate(True), stream_rate(False)
    # the bare stage batch: eval forward of the generator on one batch of that many slices (HIP events, inputs resident)
    bb = synth.to_model_inputs(synth.make_batch(max(nz, 1), 256, seed=3))
    a = [bb['real_A'].to(dev), bb['mask'].to(dev), (1 - bb['CAM']).to(dev), bb['slice_ratio'].to(dev)]
    for _ in range(2):
        net.run_forward(*a, training=False, per_sample_mask=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        net.run_forward(*a, training=False, per_sample_mask=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return {'workload': 'eval_3d_sagittal_twostage: 256x256x64 synthetic volume, upper -> lower -> target synthesis, each stage one batched launch',
            'ms_per_volume': round(dt * 1e3, 2), 'volumes_per_s': round(1.0 / dt, 2), 'slices_with_output': nz,
            'pipelined': {'ms_per_volume': round(dt_pipe * 1e3, 2), 'volumes_per_s': round(1.0 / dt_pipe, 2),
                          'ms_per_volume_pinned_views': round(dt_pipe_views * 1e3, 2),
                          'what': 'steady state of infer.process_volumes over 10 volumes (first two not counted): one pinned host copy per volume, upload / z-extent scan / '
                                  'float32 conversion / output volumes on the device, transfers of neighbouring volumes overlapped with the stages; '
                                  'pinned_views = outputs handed out as views of the pinned download buffers instead of fresh arrays'},
            'slice_stages_per_s': round(3 * nz / dt, 1), 'stage_batch_forward_ms': round(ms, 3),
            'stage_batch_slices_per_s': round(max(nz, 1) / ms * 1e3, 1),
            'stage_batch_tflops': round(GFLOP_G_FWD * max(nz, 1) / ms, 1), 'includes': 'host slicing, PCIe in/out, device preparation, 3 generator stages'}


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))           # parent: spawn the ranks, never touch the GPU
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but %d rank(s) are running (WORLD_SIZE)' % (args.gpus, world))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist
    os.environ['HV_PRECISION'] = args.precision
    if torch.cuda.device_count() and local_rank >= torch.cuda.device_count():
        local_rank %= torch.cuda.device_count()      # rehearsal of N ranks on fewer devices (HV_DDP_BACKEND=gloo)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    import hvgan  # noqa: F401
    from hvgan import synth, ddp, profiler, engine
    from hvgan.models.pix2pix_model import Pix2PixModel

    torch.manual_seed(1234)                       # same initial weights on every rank (and rank 0's are broadcast)
    opt = make_opt(args.precision)
    opt.gpu_ids = [local_rank]
    model = Pix2PixModel(opt)                     # joins the RCCL job described by the environment (ddp.init_from_env)
    model.setup(opt)
    model.strict_graph = True                     # a failed hipGraph capture is an error here, never a silent eager fallback
    if world > 1 and not (dist.is_initialized() and dist.get_world_size() == world):
        raise SystemExit('bench.py: process group not initialised for %d ranks' % world)
    batch = synth.make_batch(args.batch, args.size, seed=1234 + rank)     # weak scaling: each rank its own 16 slices
    model.set_input(batch)                        # inputs resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = model.optimize_parameters
    if args.serial:
        engine.SERIAL = True
    if args.no_graph:
        model.use_graph = False
    # W untimed warm-up steps; the step graphs are captured after the model's first eager steps, so a warm-up shorter than
    # that is topped up (untimed) to keep the capture out of the timed region
    for _ in range(max(args.warmup, model.GRAPH_WARMUP + 1 if model.use_graph else 1)):
        step()
    barrier()
    if model.use_graph and not model._graphs:
        raise SystemExit('bench.py: the step was not captured as hipGraphs')
    prof = profiler.KernelTimer()
    prof.calibrate()                  # what an event pair reads around NOTHING on this box: taken off every kernel's event duration below
    serial0 = engine.SERIAL
    engine.SERIAL = True              # per-kernel HIP-event timing: one stream, the kernel has the GPU to itself
    prof.enable()                     # untimed eager survey step: time every conv launch, pick the dominant kernel class
    step()
    dom = prof.dominant()
    dom_mfma = prof.dominant_mfma(MFMA_PEAK_TFLOPS[args.precision])      # the largest MFMA-bound instantiation (the PatchGAN 4x4 layers), timed beside it
    prof.disable()
    engine.SERIAL = serial0
    # the timed region: K steps (graph replays unless --no-graph) between barrier + synchronize on both sides -- run three times back to back,
    # the MEDIAN region is reported (all three are on the record: `regions_ms_per_step`)
    region_dt = []
    for _ in range(3):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        region_dt.append(time.perf_counter() - t0)
    # roofline leg: the same K steps once more, launched eagerly on ONE stream with HIP events around the dominant
    # kernel's launches (a captured graph cannot carry timing events, and with other streams busy an event pair would
    # also time the wait for free CUs); `bench.py --serial` under rocprofv3 gives the matching per-kernel averages.
    # The same eager single-stream steps carry the event pair around the refinement generator's forward.
    engine.SERIAL = True
    prof.enable(only=(list(dom[1]) + (list(dom_mfma[1]) if dom_mfma else [])) if dom else None)
    model.netG.time_fine = []
    for _ in range(args.steps):
        step()
    barrier()
    prof.disable()
    fine = model.netG.time_fine
    model.netG.time_fine = None
    engine.SERIAL = serial0
    # the same event pair with the step's real stream structure (the refinement generator's dilated-conv branch and attention branch on two
    # streams, as inside the captured graphs): a few eager steps with the side streams on
    fine2 = None
    if not args.serial:
        ug, model.use_graph = model.use_graph, False
        model.netG.time_fine = []
        for _ in range(5):
            step()
        barrier()
        fine2 = model.netG.time_fine
        model.netG.time_fine = None
        model.use_graph = ug
    # ... and the refinement generator's forward ALONE as the product launches it: one captured hipGraph (both branches on their two streams),
    # replayed with nothing else on the GPU.  The two event pairs above are taken around eager launches (one Python / ctypes call per kernel:
    # the host can be the bound); this one is the device time of the ~65 kernels
    fine_graph = None
    if world == 1 and not args.serial and model.use_graph and getattr(model, '_gplan', None) is not None:
        try:
            torch.cuda.synchronize()
            cam_t = model._buf('cam_temp', model.CAM)
            g = model.netG.fine_forward_graph(model._gplan, model.real_A, model.mask, model.slice_ratio)
            for _ in range(3):
                g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nrep = 50
            e0.record()
            for _ in range(nrep):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            fine_graph = e0.elapsed_time(e1) / nrep
            del g, cam_t
        except Exception as exc:       # noqa: BLE001 -- a diagnostic leg: the headline numbers do not depend on it
            sys.stderr.write('bench.py: fine-generator graph replay leg skipped (%s)\n' % str(exc).splitlines()[0])
    if world > 1:      # MAX over ranks, region by region
        t = torch.tensor(region_dt, device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        region_dt = [float(v) for v in t.tolist()]
    dt = sorted(region_dt)[1]
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt

    out = {
        'metric': 'sagittal slices/sec (train step, G+D) at %dx%d bs=%d' % (args.size, args.size, args.batch),
        'value': round(value, 2), 'unit': 'slices/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f16' if args.precision == 'fp16' else 'f32', 'data': 'synthetic',
        'config': {'workload': 'two-stage coarse+refine generator + 3x PatchGAN D train step, %dx%d, per-GPU bs=%d, norm=batch, vanilla GAN'
                               % (args.size, args.size, args.batch),
                   'global_batch': world * args.batch, 'precision': engine.precision_note(args.precision), 'parallelism': 'dp%d' % world},
        'achieved_tflops': round(GFLOP_PER_SLICE * value / 1e3, 2),
        'regions_ms_per_step': [round(r / args.steps * 1e3, 3) for r in region_dt],
        # what the collective layer actually saw (an N-GPU record must show N ranks behind RCCL)
        'comm': {'backend': (dist.get_backend() if dist.is_initialized() else None), 'world_size': (dist.get_world_size() if dist.is_initialized() else 1),
                 'grad_exchange': (({'captured': 'two ncclAllReduce (ncclAvg) captured inside the step graph on its main branch: the three discriminators\' gradient arena where their streams join, G\'s before its Adam step',
                                     'overlapped': 'flat all-reduce (ncclAvg) per network captured inside the step graph, chained D_1 -> D_2 -> D_3 -> G by events: D_k\'s beside the other discriminators\' passes, G\'s before its Adam step'}
                                    .get(model.dp_schedule, '?') if getattr(model, '_inline_exchange', False) else 'flat all-reduce (mean) between the step\'s three graphs on the exchange stream')
                                   if model.grad_sync.active() else 'none (single rank)'),
                 'transport': (('RCCL through our own communicator (ddp.RcclComm)' if model.grad_sync.capturable() else 'torch.distributed (%s)' % dist.get_backend())
                               if model.grad_sync.active() else None),
                 'dp_schedule': (model.dp_schedule if model.grad_sync.active() else None), 'capture_error': getattr(model, 'dp_capture_error', None),
                 # both schedules run on this job's first batch before the warm-up: per schedule the slowest rank's ms/step, whether every rank ended with
                 # the same weights (exact checksum), the error text if it failed; `chosen` is what the timed region ran
                 'preflight': getattr(model, 'dp_preflight_record', None)},
    }
    if rank == 0:
        out['roofline'] = prof.roofline(args.precision, MFMA_PEAK_TFLOPS[args.precision], name=dom[0] if dom else None)
        if out['roofline'] is None:
            raise SystemExit('bench.py: the roofline leg timed no kernel launch')
        # the dominant instantiation by total time is an HBM-side generator kernel since round 2; the largest MFMA-bound one (same events, same
        # steps) keeps the matrix-core kernels of the step on the record
        if dom_mfma and dom_mfma[0] != dom[0]:
            out['roofline_mfma'] = prof.roofline(args.precision, MFMA_PEAK_TFLOPS[args.precision], name=dom_mfma[0])
        # HBM bytes per launch from the committed PMC passes of this round (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate
        # passes over a serial bench run; mean per launch of this instantiation)
        pdir = os.path.join(ROOT, 'profiles')

        import glob
        traffic_files = sorted(glob.glob(os.path.join(pdir, 'r*_traffic_step.json')), reverse=True)      # newest round / tag first

        def add_traffic(rec):
            for fp in traffic_files:
                try:
                    e = json.load(open(fp))['kernels'].get(rec['kernel'])
                except (OSError, ValueError, KeyError):
                    e = None
                if e:
                    rec['traffic'] = int(e['traffic_bytes'])
                    rec['traffic_source'] = 'profiles/%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, serial bench run)' % os.path.basename(fp)
                    break
        add_traffic(out['roofline'])
        if out.get('roofline_mfma'):
            add_traffic(out['roofline_mfma'])
        out['roofline']['timed_in'] = ('eager single-stream re-run of the K steps after the timed region (HIP events on the launch stream); avg_us is net of the '
                                       'event pair\'s own reading around nothing (event_pair_overhead_us, measured in this run), avg_us_events is the raw reading')
        # the whole step against both roofs: algorithmic FLOPs of the step and the HBM bytes the committed PMC passes measured per step
        step_rec = {'achieved_tflops': round(GFLOP_PER_SLICE * args.batch * (args.size / 256.0) ** 2 / ms, 1)}
        step_rec['frac_mfma'] = round(step_rec['achieved_tflops'] / MFMA_PEAK_TFLOPS[args.precision], 4)
        for fp in traffic_files:
            try:
                doc = json.load(open(fp))
                tot = doc.get('step_total_bytes')
                if not tot:      # (files written before round 4: steps of the profiled run = launches of the once-per-step compositing kernel)
                    once = doc['kernels'].get('post_generator_kernel', {}).get('launches')
                    tot = int(sum(v['traffic_bytes'] * v['launches'] for v in doc['kernels'].values()) / once) if once else None
            except (OSError, ValueError, KeyError):
                tot = None
            if tot and args.size == 256 and args.batch == 16 and args.precision == 'fp16':
                step_rec.update(hbm_gb_per_step=round(tot / 1e9, 2), hbm_gbs=round(tot / 1e9 / (ms * 1e-3), 1), frac_hbm=round(tot / 1e9 / (ms * 1e-3) / profiler.HBM_PEAK_GBS, 4),
                                traffic_source='profiles/%s (sum over every kernel of the step, PMC passes)' % os.path.basename(fp))
                break
        out['step_roofline'] = step_rec
        if fine:
            torch.cuda.synchronize()
            fms = sum(s.elapsed_time(e) for s, e in fine) / len(fine)
            tf = GFLOP_FINE_FWD * args.batch * (args.size / 256.0) ** 2 / fms
            out['fine_generator_forward'] = {'ms': round(fms, 3), 'gflop': round(GFLOP_FINE_FWD * args.batch * (args.size / 256.0) ** 2, 1),
                                             'tflops': round(tf, 1), 'frac_of_mfma_peak': round(tf / MFMA_PEAK_TFLOPS[args.precision], 4),
                                             'algorithmic_mb': round(MB_FINE_FWD * args.batch * (args.size / 256.0) ** 2 * (1 if args.precision == 'fp16' else 2), 1),
                                             'frac_of_hbm_peak': round(MB_FINE_FWD * args.batch * (args.size / 256.0) ** 2 * (1 if args.precision == 'fp16' else 2) / 1e3 / fms / profiler.HBM_PEAK_GBS * 1e3, 4),
                                             'timed_in': 'the same eager single-stream steps: HIP events around FineGenerator (training forward, '
                                                         'both branches + contextual attention in line), mean of %d' % len(fine)}
        if fine2:
            f2 = sum(s_.elapsed_time(e_) for s_, e_ in fine2) / len(fine2)
            tf2 = GFLOP_FINE_FWD * args.batch * (args.size / 256.0) ** 2 / f2
            out['fine_generator_forward']['two_streams'] = {'ms': round(f2, 3), 'tflops': round(tf2, 1), 'frac_of_mfma_peak': round(tf2 / MFMA_PEAK_TFLOPS[args.precision], 4),
                                                            'timed_in': 'eager steps with the step\'s own streams (the two branches concurrent, discriminator streams busy beside them), mean of %d' % len(fine2)}
        if fine_graph and 'fine_generator_forward' in out:
            tfg = GFLOP_FINE_FWD * args.batch * (args.size / 256.0) ** 2 / fine_graph
            out['fine_generator_forward']['graph_replay'] = {'ms': round(fine_graph, 3), 'tflops': round(tfg, 1), 'frac_of_mfma_peak': round(tfg / MFMA_PEAK_TFLOPS[args.precision], 4),
                                                             'frac_of_hbm_peak': round(MB_FINE_FWD * args.batch * (args.size / 256.0) ** 2 / 1e3 / fine_graph / profiler.HBM_PEAK_GBS * 1e3, 4),
                                                             'timed_in': 'the refinement generator\'s training forward alone as one captured hipGraph (two branch streams), mean of 50 back-to-back replays, GPU otherwise idle'}
        ngraphs = len(model._graphs or ())
        out['config']['launch'] = ('hipGraph replay (%d graphs/step)' % ngraphs if model.use_graph else 'eager') + \
                                  (', one stream' if args.serial else ', %d streams' % (5 if (world > 1 or model.grad_sync.active()) else 4))
        out['losses'] = {k: round(v, 4) for k, v in model.get_current_losses().items()}
        if world == 1 and not args.no_inference and args.size == 256:
            out['inference'] = inference_record(dev, args.precision)
        if world == 1 and not args.no_extra and args.precision == 'fp16':
            out['device_loader'] = device_loader_record(model, args, dev)
            out['fp32'] = fp32_record(args, dev, local_rank)
            if args.size == 256:
                out['config5'] = config5_record(args, dev, local_rank)
                out['config5']['per_pixel_rate_vs_256'] = round(out['config5']['pixels_per_s'] / (value * 256 * 256 / 1e6), 3)
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(model, args.size, 1234)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
