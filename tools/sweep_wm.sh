#!/bin/bash
# wave arrangement of conv_halo2's 4x4 stride-1 instantiations (HV_HALO2_WM: bit 0 -> 64-channel blocks as 2x2 waves, bit 1 -> 128-channel blocks)
b="python tools/bench_conv.py"
for knob in "HV_HALO2_WM=0" "HV_HALO2_WM=1" "HV_HALO2_WM=2"; do
  echo "== $knob"
  env $knob $b 16 32 32 256 512 4 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 31 31 512 256 4 1 1 1 30 2>/dev/null | tail -1
  env $knob $b 16 64 64 64 128 4 1 1 0 30 2>/dev/null | tail -1
  env $knob $b 16 63 63 128 64 4 1 1 1 30 2>/dev/null | tail -1
done
