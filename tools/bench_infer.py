"""Inference throughput (BASELINE config #4, secondary metric): the three-stage iterative synthesis of one straightened
256 x 256 x 64 volume (reference eval_3d_sagittal_twostage.py:186-234) with each stage batched over all z-slices, and the bare
eval-mode generator forward at bs = 1 / 16 / 64.  Synthetic volume and random-init weights (no checkpoints in the container)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('HV_PRECISION', 'fp16')
import hvgan
from hvgan import synth, infer
from hvgan.models.inpaint_networks import Generator

torch.manual_seed(0)
dev = torch.device('cuda:0')
net = Generator({'input_dim': 1, 'ngf': 16}, True)
net.fine_generator.fc_height.bias.data.fill_(0.4)
net.fine_generator.fc_height.weight.data.mul_(1e-2)
net.cuda().train()
b = synth.to_model_inputs(synth.make_batch(4, 256, seed=1))
for _ in range(3):     # settle the spectral-norm power iteration of the random weights
    net.run_forward(b['real_A'].to(dev), b['mask'].to(dev), (1 - b['CAM']).to(dev), b['slice_ratio'].to(dev), training=True)
net.eval()

ct, label, cam = synth.make_volume(nz=64, size=256, seed=2)
for _ in range(2):
    infer.process_volume(net, ct, label, cam * 255, 20, dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 5
for _ in range(N):
    out_ct, out_seg = infer.process_volume(net, ct, label, cam * 255, 20, dev)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / N
nz = int((out_seg.reshape(-1, out_seg.shape[2]) != 0).any(axis=0).sum())
print('process_volume 256x256x64: %.1f ms per volume (host pre-processing included), %d slices with output -> %.0f slice-stages/s'
      % (dt * 1e3, nz, 3 * nz / dt))

for B in (1, 16, 64):
    bb = synth.to_model_inputs(synth.make_batch(B, 256, seed=3))
    args = [bb['real_A'].to(dev), bb['mask'].to(dev), (1 - bb['CAM']).to(dev), bb['slice_ratio'].to(dev)]
    for _ in range(3):
        net.run_forward(*args, training=False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 20
    for _ in range(n):
        net.run_forward(*args, training=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('eval forward bs=%d: %.2f ms -> %.0f slices/s (%.1f TFLOP/s at 17.56 GFLOP/slice)' % (B, ms, B / ms * 1e3, 17.56 * B / ms))

for B in (1, 16):     # the nn.Module call of the eval scripts: hipGraph replay per input shape (+ copies of the 6 outputs)
    bb = synth.to_model_inputs(synth.make_batch(B, 256, seed=3))
    args = [bb['real_A'].to(dev), bb['mask'].to(dev), (1 - bb['CAM']).to(dev), bb['slice_ratio'].to(dev)]
    with torch.no_grad():
        for _ in range(3):
            net(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50
        for _ in range(n):
            net(*args)
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print('netG(...) eval, no_grad, bs=%d (graph replay): %.2f ms per call -> %.0f slices/s' % (B, ms, B / ms * 1e3))
