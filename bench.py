#!/usr/bin/env python
"""bench.py -- sagittal slices/sec of one full HealthiVert-GAN train step (G + 3xD) at 256x256, per-GPU bs=16.

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

One "step" = Pix2PixModel.optimize_parameters on one synthetic batch already resident in HBM: generator forward,
three discriminator updates (fake + real pass each, Adam), generator backward through D_1/D_3 + losses, Adam.
Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel, measured live with HIP
events on the launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores, rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_SLICE = 195.9          # SURVEY.md section 8d / BASELINE.md section 2: whole train step, per slice
MFMA_PEAK_TFLOPS = {'fp16': 2500.0, 'fp32': 157.3}   # MI355X dense peaks, /opt/skills/guides/MI355X_MICROARCH.md


def make_opt(precision):
    return Namespace(gpu_ids=[0], isTrain=True, checkpoints_dir='/tmp/hv_bench', name='bench', preprocess='none', input_nc=1,
                     output_nc=1, ngf=64, ndf=64, netD='basic', netG='unet_256', n_layers_D=3, norm='batch', init_type='normal',
                     init_gain=0.02, no_dropout=True, gan_mode='vanilla', lr=2e-4, beta1=0.5, lambda_L1=200.0, direction='BtoA',
                     lr_policy='linear', epoch_count=1, n_epochs=100, n_epochs_decay=100, continue_train=False, load_iter=0,
                     epoch='latest', verbose=False, hv_precision=precision)


def cpu_baseline(model, batch_size, size, seed):
    """The CPU oracle (oracle/restate.py, kind 'port') on the same weights and the same synthetic batch: ONE full
    train step at the bench batch size (bounded sample: ~20-30 s of CPU work)."""
    from hvgan import synth
    from oracle import restate as R
    sd_g = {k: v.detach().cpu() for k, v in model.netG.state_dict().items()}
    sd_d = [{k: v.detach().cpu() for k, v in getattr(model, 'netD_%d' % k).state_dict().items()} for k in (1, 2, 3)]
    st = R.StepState(sd_g, sd_d, lr=2e-4, beta1=0.5, norm='batch', gan_mode='vanilla', lambda_l1=200.0)
    batch = synth.to_model_inputs(synth.make_batch(batch_size, size, seed=seed))
    t0 = time.time()
    R.pix2pix_step(st, batch)
    dt = time.time() - t0
    return dict(value=batch_size / dt, unit='slices/s', cores=torch.get_num_threads(), kind='port',
                sample='1 full train step (G + 3xD) at bs=%d, %dx%d, fp32, CPU oracle oracle/restate.py' % (batch_size, size, size),
                seconds=round(dt, 2))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=16, help='per-GPU batch (slices)')
    ap.add_argument('--size', type=int, default=256)
    ap.add_argument('--precision', default=os.environ.get('HV_PRECISION', 'fp16'), choices=['fp16', 'fp32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--serial', action='store_true', help='one HIP stream for the whole run (profiling: per-kernel durations without stream-level overlap)')
    ap.add_argument('--no-graph', action='store_true', help='launch kernels eagerly instead of replaying a captured hipGraph')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus != world and world > 1:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    os.environ['HV_PRECISION'] = args.precision
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)

    import hvgan
    from hvgan import synth, ddp, profiler
    from hvgan.models.pix2pix_model import Pix2PixModel

    torch.manual_seed(1234)                       # same initial weights on every rank
    opt = make_opt(args.precision)
    opt.gpu_ids = [local_rank]
    model = Pix2PixModel(opt)
    model.setup(opt)
    ddp.broadcast_parameters([model.netG, model.netD_1, model.netD_2, model.netD_3])
    batch = synth.make_batch(args.batch, args.size, seed=1234 + rank)     # weak scaling: each rank its own 16 slices
    model.set_input(batch)                        # inputs resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    step = model.optimize_parameters
    from hvgan import engine
    if args.serial:
        engine.SERIAL = True
    if args.no_graph:
        model.use_graph = False
    # W untimed warm-up steps; the step graph is captured after the model's first eager steps, so a warm-up shorter than
    # that is topped up (untimed) to keep the capture out of the timed region
    for _ in range(max(args.warmup, model.GRAPH_WARMUP + 1 if model.use_graph else 1)):
        step()
    barrier()
    prof = profiler.KernelTimer()
    serial0 = engine.SERIAL
    engine.SERIAL = True              # per-kernel HIP-event timing: one stream, the kernel has the GPU to itself
    prof.enable()                     # untimed eager survey step: time every conv launch, pick the dominant kernel class
    step()
    dom = prof.dominant()
    prof.disable()
    engine.SERIAL = serial0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):       # the timed region: K steps (graph replays unless --no-graph)
        step()
    barrier()
    dt = time.perf_counter() - t0
    # roofline leg: the same K steps once more, launched eagerly on ONE stream with HIP events around the dominant
    # kernel's launches (a captured graph cannot carry timing events, and with other streams busy an event pair would
    # also time the wait for free CUs); `bench.py --serial` under rocprofv3 gives the matching per-kernel averages
    engine.SERIAL = True
    prof.enable(only=dom[1] if dom else None)
    for _ in range(args.steps):
        step()
    barrier()
    prof.disable()
    engine.SERIAL = serial0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = world * args.batch * args.steps / dt

    out = {
        'metric': 'sagittal slices/sec (train step, G+D) at %dx%d bs=%d' % (args.size, args.size, args.batch),
        'value': round(value, 2), 'unit': 'slices/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(ms, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f16' if args.precision == 'fp16' else 'f32', 'data': 'synthetic',
        'config': {'workload': 'two-stage coarse+refine generator + 3x PatchGAN D train step, %dx%d, per-GPU bs=%d, norm=batch, vanilla GAN'
                               % (args.size, args.size, args.batch),
                   'global_batch': world * args.batch, 'precision': 'fp16 MFMA operands / fp32 accumulate, fp32 storage'
                   if args.precision == 'fp16' else 'fp32 MFMA', 'parallelism': 'dp%d' % world},
        'achieved_tflops': round(GFLOP_PER_SLICE * value / 1e3, 2),
    }
    if rank == 0:
        out['roofline'] = prof.roofline(args.precision, MFMA_PEAK_TFLOPS[args.precision])
        # HBM bytes per launch from the committed PMC passes: per layer shape where that kernel was profiled alone
        # (profiles/r01_traffic.json), else the per-instantiation mean over the whole step (profiles/r01_traffic_step.json)
        pdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles')
        try:
            tr = json.load(open(os.path.join(pdir, 'r01_traffic.json')))['kernels']
            ents = [(tr.get(sh['shape']), sh['launches']) for sh in out['roofline']['shapes']]
            if ents and all(e for e, _ in ents):      # launch-weighted mean over the layer shapes this instantiation serves
                out['roofline']['traffic'] = int(sum(e['traffic_bytes'] * n for e, n in ents) / sum(n for _, n in ents))
                out['roofline']['traffic_source'] = 'profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)'
            else:
                e = json.load(open(os.path.join(pdir, 'r01_traffic_step.json')))['kernels'].get(out['roofline']['kernel'])
                if e:
                    out['roofline']['traffic'] = int(e['traffic_bytes'])
                    out['roofline']['traffic_source'] = ('profiles/r01_traffic_step.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate '
                                                         'passes over a serial bench run; mean per launch of this instantiation)')
        except (OSError, ValueError, KeyError):
            pass
        out['roofline']['timed_in'] = 'eager single-stream re-run of the K steps after the timed region (HIP events on the launch stream)'
        out['config']['launch'] = ('hipGraph replay (3 graphs/step)' if model.use_graph else 'eager') + (', one stream' if args.serial else ', 4 streams')
        out['losses'] = {k: round(v, 4) for k, v in model.get_current_losses().items()}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(model, args.batch, args.size, 1234)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
