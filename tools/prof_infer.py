"""cProfile of infer.process_volume (host side)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('HV_PRECISION', 'fp16')
import hvgan
from hvgan import synth, infer
from hvgan.models.inpaint_networks import Generator
torch.manual_seed(0)
dev = torch.device('cuda:0')
net = Generator({'input_dim': 1, 'ngf': 16}, True)
net.fine_generator.fc_height.bias.data.fill_(0.4); net.fine_generator.fc_height.weight.data.mul_(1e-2)
net.cuda().eval()
ct, label, cam = synth.make_volume(nz=64, size=256, seed=2)
for _ in range(2):
    infer.process_volume(net, ct, label, cam * 255, 20, dev)
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(3):
    infer.process_volume(net, ct, label, cam * 255, 20, dev)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
