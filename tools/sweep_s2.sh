#!/bin/bash
# the PatchGAN stride-2 layers under the kernel-choice knobs (one gpurun call: same device)
b="python tools/bench_conv.py"
for knob in "X=0" "HV_HALO_BN=64" "HV_HALO_BN=128" "HV_HALO2_S2F=1"; do
  echo "== $knob"
  env $knob $b 16 256 256 64 64 3 1 1 0 5 > /dev/null 2>&1
  env $knob $b 16 128 128 64 128 4 2 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 128 256 4 2 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 128 64 4 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 32 32 256 128 4 2 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 32 32 256 512 4 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 31 31 512 256 4 1 1 1 30 2>/dev/null | tail -1
done
